#!/bin/bash
# Two-phase rule away from the bench shape: k = 2:16 x 32 restarts, 200 iterations, fixed budget;
# automatic schedule vs forced one-phase packed-VALU sweep (NMFK_HYB=0) vs forced two-phase (NMFK_HYB_PHASES=1).
cat > /tmp/shape_bench.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import nmfk_jl_amd as N
n, m, iters, kmax, R = (int(v) for v in sys.argv[1:6])
ctx = N.Context(0)
X = ctx.fill_uniform(1, 0, n * m).reshape(m, n).T
ctx.set_X(X)
ks = list(range(2, kmax + 1))
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
ctx.set_profiling(True)
res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
p = ctx.get_profile()["mu_loop"]
info = ctx.last_sweep_info()
print("%6d x %-5d k=2:%d x %d  %-28s MU loop %8.2f ms / %d it   phases=%d mfma_units=%d" % (
    n, m, kmax, R, os.environ.get("TAG", ""), p["ms"], iters, info["phases"], info["mfma_group_units"]))
PY
SHAPES=("8192 512 200" "2048 2048 200" "65536 256 60" "1024 128 400" "4096 64 400" "512 8192 200")
for shape in "${SHAPES[@]}"; do
  for kr in ${KRS:-"16 32" "12 32" "16 16"}; do
    TAG=auto python /tmp/shape_bench.py $shape $kr
    TAG=NMFK_HYB=0 NMFK_HYB=0 python /tmp/shape_bench.py $shape $kr
    TAG=NMFK_HYB_PHASES=1 NMFK_HYB=1 NMFK_HYB_MINK=9 NMFK_HYB_PHASES=1 python /tmp/shape_bench.py $shape $kr
  done
done
