#!/bin/bash
# Known hazard: the fp32 merged packed-VALU kernel (checker process: must reproduce its first result bit for
# bit) beside SYNTHETIC neighbours that keep one hardware unit busy each (tools/hazard/burner.hip).
cd $(dirname $0)/../..
for b in burner pk_victim cu_map uniform_vload; do [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scratch/$b.hip -o scratch/$b 2>/dev/null; done
# (to see the hazard again: build the library from a commit before the broadcast-first operand rule, or with the rule reverted)
for spec in "burner 0" "burner_vf 0" "burner 1" "burner_vf 1" "burner 2" "burner 3" "burner 4" "burner 5"; do
  set -- $spec
  echo "== $1 mode $2"
  timeout -k 5 60 ./scratch/$1 $2 28 &
  BURN=$!
  sleep 3
  NMFK_HYB=0 NMFK_MERGE=1 KS=2,3,5 timeout -k 5 60 python tools/hazard/dbg_sidebyside.py ${REPS:-150} 4 2>&1 | tail -1
  wait $BURN
done
