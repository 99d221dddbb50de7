#!/bin/bash
# Two PROCESSES on one GPU: a burner running MFMA-group sweeps, and a checker running merged packed-VALU sweeps (fp32,
# NMFK_HYB=0) that must reproduce its first result bit for bit.
cd $(dirname $0)/..
KS=13,16,9,12 timeout -k 5 120 python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, nmfk_jl_amd as NMFk, nmfk_oracle as oracle
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx = NMFk.Context(0); ctx.set_X(X)
ks = [13, 16, 9, 12]; R = 8
seeds = np.array([[NMFk.run_seed(5, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
t0 = time.time(); nsw = 0
while time.time() - t0 < 60:
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=40, maxbaditers=10 ** 9); nsw += 1
print("burner: sweeps", nsw, ctx.last_sweep_info(), flush=True)
PY
BURN=$!
sleep 8
NMFK_HYB=0 KS=${CHK_KS:-2,3,5} timeout -k 5 100 python scripts/dbg_sidebyside.py ${REPS:-400} ${RCHK:-4} 2>&1 | tail -2
wait $BURN
