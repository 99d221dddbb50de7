#!/usr/bin/env python3
"""H half-step of the bench sweep with its loop range split over S workgroups (NMFK_FORCE_SH, experiment): the 960 workgroups of
S = 1 run as one round of 512 long ones (k >= 9) and one of 448 shorter ones on the 512 slots of the chip."""
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
ctx.set_profiling(True)
for rep in range(2):
    for sh in ("0", "2", "3", "4"):
        os.environ["NMFK_FORCE_SH"] = sh
        t = time.perf_counter()
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=600)
        dt = time.perf_counter() - t
        prof = ctx.get_profile()
        hs = {k: round(v["ms"] / max(v["launches"], 1), 4) for k, v in prof.items() if k.startswith(("h_step", "w_step"))}
        print(f"NMFK_FORCE_SH={sh}: {dt / 600 * 1e3:.4f} ms per iteration; sampled launches {hs}; objvalue k=16 r=0 {res[16]['objvalue'][0]:.6f}", flush=True)
