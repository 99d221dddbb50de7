#!/usr/bin/env python3
"""Shapes whose short dimension is not a multiple of 64: the W half-step in its resident form (round 4: any loop length) against
the streaming form (NMFK_HYB_RES=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ctx = N.Context(0)
for (n, m) in ((20000, 1000), (8192, 500), (50000, 300)):
    X = ctx.fill_uniform(5, 0, n * m).reshape(m, n).T
    ctx.set_X(X)
    for ks, R in ((list(range(2, 17)), 32), (list(range(2, 17)), 10), (list(range(2, 6)), 10), ([4], 64), ([8], 32)):
        seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
        out = {}
        for mode in ("1", "0"):
            os.environ["NMFK_HYB_RES"] = mode
            os.environ["NMFK_HYB"] = "1"; os.environ["NMFK_HYB_MINK"] = "2"; os.environ["NMFK_HYB_PHASES"] = "1"
            ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
            best = 1e9
            for rep in range(2):
                t = time.perf_counter()
                res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=200, maxbaditers=10 ** 9)
                best = min(best, time.perf_counter() - t)
            out[mode] = (best / 200 * 1e3, float(res[ks[-1]]["objvalue"][0]))
        os.environ["NMFK_HYB"] = "0"
        t = time.perf_counter()
        ctx.mu_sweep(ks, R, seeds=seeds, maxiter=200, maxbaditers=10 ** 9)
        valu = (time.perf_counter() - t) / 200 * 1e3
        print(f"{n} x {m}  k = {ks[0]}..{ks[-1]} x {R}: resident {out['1'][0]:.4f} ms per iteration, streaming {out['0'][0]:.4f}, packed-VALU launches {valu:.4f}; objvalue {out['1'][1]:.6f} / {out['0'][1]:.6f}", flush=True)
