#!/usr/bin/env python3
"""BASELINE configs[4]: planted rank-48 + noise, 65536 x 2048, k = 64, nruns restarts, reference stop rule.
usage: bench_cfg5.py [nruns=64] [maxiter=10000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
maxiter = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
n, m, k, k0 = 65536, 2048, 64, 48
ctx = N.Context(0)
W0 = ctx.fill_uniform(4, 0, n * k0).reshape(k0, n).T
H0 = ctx.fill_uniform(5, 0, k0 * m).reshape(m, k0).T
X = (W0 @ H0 + 0.01 * ctx.fill_uniform(6, 0, n * m).reshape(m, n).T).astype(np.float32)
del W0, H0
ctx.set_X(X)
ctx.set_profiling(True)
t = time.perf_counter()
W, H, fit, rob, aic, det = N.execute(X, k, R, load=False, save=False, quiet=True, seed=4, ctx=ctx, maxiter=maxiter,
                                     return_details=True)
dt = time.perf_counter() - t
p = ctx.get_profile()
loop = p.get("mu_loop", {"ms": 0, "flops": 0})
its = det["iters"] if det and "iters" in det else None
print(f"cfg5 k={k} nruns={R}: {dt:.1f} s = {R / dt:.3f} factorizations/s; fit {fit:.4g} robustness {rob:.3f}; "
      f"MU loop {loop['ms'] / 1e3:.1f} s, {loop['flops'] / max(loop['ms'], 1e-9) / 1e9:.1f} TFLOP/s; "
      f"iterations {None if its is None else (int(np.min(its)), int(np.max(its)))}")
