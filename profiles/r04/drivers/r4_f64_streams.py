#!/usr/bin/env python3
"""fp64 compute (per-rank launch groups): 8 against 16 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ctx = N.Context(0)
n, m = 8192, 512
X = ctx.fill_uniform(5, 0, n * m).reshape(m, n).T
ctx.set_X(X)
for ks, R in ((list(range(2, 17)), 8), (list(range(2, 17)), 32), (list(range(17, 33)), 8)):
    seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    out = []
    for st in ("8", "16", "8", "16"):
        os.environ["NMFK_STREAMS"] = st
        ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9, compute=N.COMPUTE_F64)
        t = time.perf_counter()
        ctx.mu_sweep(ks, R, seeds=seeds, maxiter=100, maxbaditers=10 ** 9, compute=N.COMPUTE_F64)
        out.append(f"streams={st} {(time.perf_counter() - t) / 100 * 1e3:.4f}")
    print(f"f64 {n} x {m} k = {ks[0]}..{ks[-1]} x {R}: " + "   ".join(out), flush=True)
