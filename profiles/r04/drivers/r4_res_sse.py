#!/usr/bin/env python3
"""Wide, short X (the H half-step in its resident form): the deferred check there (round 4, late) against the plain order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ctx = N.Context(0)
for (n, m) in ((512, 8192), (1000, 4000)):
    X = ctx.fill_uniform(5, 0, n * m).reshape(m, n).T
    ctx.set_X(X)
    ks, R = list(range(2, 17)), 32
    seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=20, maxbaditers=10 ** 9)
    for mode in ("0", "1", "0", "1"):
        os.environ["NMFK_DEFER_OBJ"] = mode
        t = time.perf_counter()
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=400, maxbaditers=10 ** 9)
        dt = time.perf_counter() - t
        info = ctx.last_sweep_info()
        print(f"{n} x {m} NMFK_DEFER_OBJ={mode}: {dt / 400 * 1e3:.4f} ms per iteration; deferred {info['deferred_checks']}, plain {info['plain_checks']}; objvalue {res[16]['objvalue'][0]:.6f}", flush=True)
