#!/bin/bash
# tools/hazard/uniform_vload beside a burner process running MFMA-group sweeps (and once alone)
cd $(dirname $0)/../..
echo "alone:"; ./tools/hazard/uniform_vload 10
timeout -k 5 120 python - <<'PY' &
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
print("burner: sweeps", nsw, flush=True)
PY
sleep 8
echo "beside the MFMA burner:"; ./tools/hazard/uniform_vload 25
wait
