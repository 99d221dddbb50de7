#!/usr/bin/env python3
"""Restarts per rank between 9 and 15 at the bench shape: the automatic schedule against the matrix-pipe group forced on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
ctx = N.Context(0)
X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
ctx.set_X(X)
forced = dict(NMFK_HYB="1", NMFK_HYB_MINK="2", NMFK_HYB_PHASES="1")
for ks, R in ((list(range(2, 17)), 8), (list(range(2, 17)), 9), (list(range(2, 17)), 10), (list(range(2, 17)), 12), (list(range(2, 17)), 15), (list(range(2, 17)), 16), (list(range(2, 6)), 10), ([3], 10), ([3], 12), ([12], 12)):
    seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=20)
    for mode in ("auto", "forced", "auto", "forced"):
        for key, val in forced.items():
            if mode == "forced":
                os.environ[key] = val
            else:
                os.environ.pop(key, None)
        t = time.perf_counter()
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=400, maxbaditers=10 ** 9)
        dt = time.perf_counter() - t
        info = ctx.last_sweep_info()
        print(f"k = {ks[0]}..{ks[-1]} x {R} restarts ({len(ks) * R} units) {mode:6s}: {dt / 400 * 1e3:.4f} ms per iteration; matrix-pipe units {info['mfma_group_units']}, launch groups {info['launch_groups']}, phases {info['phases']}", flush=True)
