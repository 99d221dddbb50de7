#!/bin/bash
# Known hazard: WHERE do the fp32 merged packed-VALU kernel (checker process) and the split-operand MFMA
# kernel (burner process) have to meet for the checker's results to change?  Both processes get CU masks (HSA_CU_MASK;
# mask bit i = CU i/8 of XCD i%8 on this part):
#   none      no masks (the known failing case)
#   cus       same XCDs (same L2s), disjoint CUs     checker bits 0-127, burner bits 128-255
#   pairs     same XCDs, even CUs vs odd CUs of every shader engine (the two CUs of a pair share the instruction and
#             scalar caches)                        (mask bit i: XCD i%8, shader engine (i/8)%4, CU i/32 -- tools/hazard/dbg_cumap.sh)
#   nows      no masks, checker without the loop split over the waves of a workgroup (NMFK_MAX_WSPLIT=1)
cd $(dirname $0)/../..
for b in burner pk_victim cu_map uniform_vload; do [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scratch/$b.hip -o scratch/$b 2>/dev/null; done
# (to see the hazard again: build the library from a commit before the broadcast-first operand rule, or with the rule reverted)
mode=${1:-none}
case $mode in
  none) CM=""; BM="";;
  cus) CM="0:0-127"; BM="0:128-255";;
  pairs) CM="0:0-31,64-95,128-159,192-223"; BM="0:32-63,96-127,160-191,224-255";;
  nows) CM=""; BM=""; export CHK_WS=1;;
esac
echo "== mode $mode"
HSA_CU_MASK=$BM timeout -k 5 100 python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, nmfk_jl_amd as NMFk, nmfk_oracle as oracle
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx = NMFk.Context(0); ctx.set_X(X)
ks = [13, 16, 9, 12]; R = 8
seeds = np.array([[NMFk.run_seed(5, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
t0 = time.time(); nsw = 0
while time.time() - t0 < 45:
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=40, maxbaditers=10 ** 9); nsw += 1
print("burner: sweeps", nsw, ctx.last_sweep_info(), flush=True)
PY
BURN=$!
sleep 8
HSA_CU_MASK=$CM NMFK_MAX_WSPLIT=${CHK_WS:-} NMFK_HYB=0 NMFK_MERGE=1 KS=2,3,5 timeout -k 5 80 python tools/hazard/dbg_sidebyside.py ${REPS:-300} 4 2>&1 | tail -2
wait $BURN
