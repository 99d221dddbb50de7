#!/bin/bash
# Which CUs do the masks of tools/hazard/dbg_cumask.sh select?  (tools/hazard/cu_map.hip)
cd $(dirname $0)/../..
for b in burner pk_victim cu_map uniform_vload; do [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scratch/$b.hip -o scratch/$b 2>/dev/null; done
echo "== no mask"; ./tools/hazard/cu_map
echo "== 0:0-127"; HSA_CU_MASK=0:0-127 ./tools/hazard/cu_map
echo "== 0:128-255"; HSA_CU_MASK=0:128-255 ./tools/hazard/cu_map
echo "== even CUs"; HSA_CU_MASK=0:0-31,64-95,128-159,192-223 ./tools/hazard/cu_map
echo "== odd CUs"; HSA_CU_MASK=0:32-63,96-127,160-191,224-255 ./tools/hazard/cu_map
echo "== 0:0-31"; HSA_CU_MASK=0:0-31 ./tools/hazard/cu_map
