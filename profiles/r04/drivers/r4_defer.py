#!/usr/bin/env python3
"""Deferred check A/B on the bench sweep (k = 2:16 x 32 restarts, 8192 x 512 U(0,1), fixed budget of 600 iterations):
NMFK_DEFER_OBJ=0 (objective launch per check) against the default (the next H half-step leaves the objective)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m, R = 8192, 512, 32
ks = list(range(2, 17))
ctx = N.Context(0)
X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
ctx.set_X(X)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, 2, seeds=seeds[:, :2], maxiter=20)
for rep in range(2):
    for mode in ("0", "1"):
        os.environ["NMFK_DEFER_OBJ"] = mode
        t = time.perf_counter()
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=600)
        dt = time.perf_counter() - t
        info = ctx.last_sweep_info()
        print(f"NMFK_DEFER_OBJ={mode}: {dt:.3f} s for 600 iterations ({dt / 600 * 1e3:.4f} ms per iteration); deferred {info['deferred_checks']}, "
              f"plain {info['plain_checks']}; objvalue k=16 r=0 {res[16]['objvalue'][0]:.6f}", flush=True)
