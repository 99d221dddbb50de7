#!/usr/bin/env python3
"""Does the automatic launch schedule of nmfk_mu_sweep pick the fastest of its alternatives?  Shapes x (ranks, restarts) x
{automatic, NMFK_HYB=0 (packed-VALU launches only), matrix-pipe group forced on for every rank}; fixed budget of 200 iterations;
flags every case where an alternative beats the automatic choice by more than 5 %."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ctx = N.Context(0)
MODES = {"auto": {}, "valu": {"NMFK_HYB": "0"}, "group": {"NMFK_HYB": "1", "NMFK_HYB_MINK": "2", "NMFK_HYB_PHASES": "1"}}
shapes = [(8192, 512), (65536, 256), (2048, 2048), (1024, 128), (512, 8192), (4096, 64), (20000, 1000), (300, 300)]
cases = [(list(range(2, 17)), 32), (list(range(2, 17)), 10), (list(range(2, 17)), 4), (list(range(2, 9)), 16), (list(range(2, 6)), 10),
         ([8], 32), ([4], 64), ([16], 10), (list(range(2, 33)), 8), (list(range(10, 21)), 10)]
if len(sys.argv) > 1:
    shapes = shapes[int(sys.argv[1])::int(sys.argv[2])]
iters = 200
for (n, m) in shapes:
    X = ctx.fill_uniform(5, 0, n * m).reshape(m, n).T
    ctx.set_X(X)
    for ks, R in cases:
        if len(ks) * R * (n + m) * max(ks) * 4 * 3 > 60e9:
            continue
        seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
        res = {}
        for mode, env in MODES.items():
            for key in ("NMFK_HYB", "NMFK_HYB_MINK", "NMFK_HYB_PHASES"):
                os.environ.pop(key, None)
            os.environ.update(env)
            ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
            best = 1e9
            for rep in range(2):
                t = time.perf_counter()
                ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
                best = min(best, time.perf_counter() - t)
            info = ctx.last_sweep_info()
            res[mode] = (best / iters * 1e3, info["mfma_group_units"], info["launch_groups"], info["phases"])
        a = res["auto"][0]
        alt = min(res["valu"][0], res["group"][0])
        flag = "  <-- LOSES %.0f %%" % (100 * (a / alt - 1)) if a > 1.05 * alt else ""
        print(f"{n:6d} x {m:5d}  k = {ks[0]:2d}..{ks[-1]:2d} x {R:2d}: auto {a:8.4f} ms (group units {res['auto'][1]}, launch groups {res['auto'][2]}, phases {res['auto'][3]})"
              f"   valu {res['valu'][0]:8.4f}   group {res['group'][0]:8.4f} (units {res['group'][1]}){flag}", flush=True)
