#!/usr/bin/env python3
"""Round 5: the launch-geometry rule of the matrix-pipe group before / after (NMFK_EXP_LEGACY_GEO=1: rounds 3-4's rule) and the cohorts,
over shapes x sweeps, fixed budget; GPU time of the MU loop per iteration (HIP events), best of 3.  usage: r5_geo_probe.py [shape0 stride]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ctx = N.Context(0)
MODES = [("old", {"NMFK_EXP_LEGACY_GEO": "1", "NMFK_COHORTS": "1"}), ("new C=1", {"NMFK_COHORTS": "1"}), ("new C=2", {"NMFK_COHORTS": "2"}),
         ("new auto", {})]
shapes = [(8192, 512), (65536, 256), (2048, 2048), (1024, 128), (512, 8192), (4096, 64), (20000, 1000), (300, 300)]
cases = [(list(range(2, 17)), 32), (list(range(2, 17)), 16), (list(range(2, 17)), 8), (list(range(2, 17)), 4), (list(range(2, 17)), 2), (list(range(2, 17)), 1),
         (list(range(2, 9)), 16), (list(range(2, 6)), 10), ([8], 32), ([8], 1), ([4], 64), ([16], 10), ([16], 2)]
if len(sys.argv) > 2:
    shapes = shapes[int(sys.argv[1])::int(sys.argv[2])]
iters = 300
for (n, m) in shapes:
    X = ctx.fill_uniform(5, 0, n * m).reshape(m, n).T
    ctx.set_X(X)
    for ks, R in cases:
        if len(ks) * R * (n + m) * max(ks) * 4 * 3 > 60e9:
            continue
        seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
        out = []
        for mode, env in MODES:
            for key in ("NMFK_EXP_LEGACY_GEO", "NMFK_COHORTS"):
                os.environ.pop(key, None)
            os.environ.update(env)
            ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
            best = 1e9
            for rep in range(3):
                ctx.set_profiling(True)
                ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
                p = ctx.get_profile()
                ctx.set_profiling(False)
                best = min(best, p["mu_loop"]["ms"] / p["mu_loop"]["launches"] / iters)
            out.append((mode + (f"[{ctx.last_sweep_info()['cohorts']}]" if mode == "new auto" else ""), best))
        base = out[0][1]
        print(f"{n:6d} x {m:5d}  k = {ks[0]:2d}..{ks[-1]:2d} x {R:2d} ({len(ks) * R:3d} units): " +
              "  ".join(f"{md} {v:.4f}" + (f" ({100 * (v / base - 1):+.0f}%)" if md != "old" else "") for md, v in out), flush=True)
