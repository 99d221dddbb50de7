#!/usr/bin/env python3
"""One wide-rank sweep for profilers: `python3 scripts/r6_wide_one.py k bn iters [n m R]` (65536 x 2048, 8 restarts by default; bn = NMFK_WIDE_BN)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
k, bn, iters = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
n, m, R = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (65536, 2048, 8)
os.environ["NMFK_WIDE_BN"] = bn
import nmfk_jl_amd as NMFk
ctx = NMFk.Context(0)
ctx.set_X(ctx.fill_uniform(4, 0, n * m).reshape(m, n).T)
seeds = np.array([[NMFk.run_seed(1, k, r) for r in range(R)]], dtype=np.uint64)
ctx.set_profiling(True)
ctx.mu_sweep([k], R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
print(k, bn, ctx.get_profile()["mu_loop"]["ms"] / iters, "ms per iteration")
