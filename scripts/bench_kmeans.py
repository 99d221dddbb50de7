#!/usr/bin/env python3
"""robustkmeans timing: rows of a W (8192 x k) into k clusters, 1000 repeats, GPU vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as N
import nmfk_oracle as oracle
ctx = N.Context(0)
for n, k in [(8192, 6), (8192, 16), (65536, 16)]:
    Wt = np.asfortranarray(ctx.fill_uniform(5, 0, n * k).reshape(n, k).T ** 3)
    ctx.robustkmeans(Wt, k, 8)
    t = time.perf_counter(); r = ctx.robustkmeans(Wt, k, 1000, seed=1); tg = time.perf_counter() - t
    reps = 4
    t = time.perf_counter(); o = oracle.robustkmeans_k(Wt, k, reps, seed=1); tc = (time.perf_counter() - t) / reps * 1000
    same = np.array_equal(o["all_costs"], r["all_costs"][:reps])
    print(f"n={n} d=k={k}: GPU 1000 repeats {tg:.3f} s (best cost {r['totalcost']:.6f}, {r['iterations']} iterations); "
          f"CPU oracle, 1 core, extrapolated from {reps} repeats: {tc:.1f} s; first {reps} repeats bit-identical: {same}")
