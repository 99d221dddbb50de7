#!/usr/bin/env python3
"""Sweeps with ranks above 16: per-rank launch groups side by side on streams (default) against one after the other
(NMFK_STREAMS=1), and the all-fp32 kernel (NMFK_WIDE2=0); 100 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ctx = N.Context(0)
MODES = {"f=0": {"NMFK_WIDE_GROUPS": "0"}, "f=1": {"NMFK_WIDE_GROUPS": "1"}, "f=2": {"NMFK_WIDE_GROUPS": "2"}, "f=4": {"NMFK_WIDE_GROUPS": "4"}, "f=1000": {"NMFK_WIDE_GROUPS": "1000"}}
for (n, m) in ((8192, 512), (20000, 1000), (1024, 128), (2048, 2048), (65536, 2048)):
    X = ctx.fill_uniform(5, 0, n * m).reshape(m, n).T
    ctx.set_X(X)
    for ks, R in ((list(range(17, 33)), 8), (list(range(17, 33)), 2), ([20, 30, 40, 50, 64], 8), (list(range(2, 41)), 4), ([24, 48], 16)):
        if len(ks) * R * (n + m) * max(ks) * 4 * 3 > 40e9:
            continue
        seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
        out = []
        for mode, env in MODES.items():
            for key in ("NMFK_STREAMS", "NMFK_WIDE2", "NMFK_WIDE_GROUPS"):
                os.environ.pop(key, None)
            os.environ.update(env)
            ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
            best = 1e9
            for rep in range(2):
                t = time.perf_counter()
                ctx.mu_sweep(ks, R, seeds=seeds, maxiter=100, maxbaditers=10 ** 9)
                best = min(best, time.perf_counter() - t)
            out.append(f"{mode} {best / 100 * 1e3:.4f}")
        info = ctx.last_sweep_info()
        print(f"{n} x {m}  k = {ks[0]}..{ks[-1]} ({len(ks)} ranks) x {R}: " + "   ".join(out) + f"   [launch groups {info['launch_groups']}]", flush=True)
