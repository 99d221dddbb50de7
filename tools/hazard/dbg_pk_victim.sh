#!/bin/bash
# tools/hazard/pk_victim.hip beside tools/hazard/burner.hip mode 0 (bf16 MFMA 16x16x32 in registers, another process), then alone
cd $(dirname $0)/../..
for b in burner pk_victim cu_map uniform_vload; do [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scratch/$b.hip -o scratch/$b 2>/dev/null; done
echo "== beside the bf16 MFMA burner"
timeout -k 5 170 ./tools/hazard/burner 0 ${SECS:-100} &
BURN=$!
sleep 3
timeout -k 5 160 ./tools/hazard/pk_victim ${REPS:-200} $ONLY
kill $BURN 2>/dev/null; wait $BURN
echo "== alone"
timeout -k 5 100 ./tools/hazard/pk_victim 60 $ONLY
