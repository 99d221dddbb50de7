#!/usr/bin/env python3
"""Round 5: launch geometries of the matrix-pipe group at few units (NMFK_EXP_GEO = "hws,hS,hres,wws,wS,wres", -1 = the rule's choice),
fixed budget at 8192 x 512, k = 2:16 x R; GPU time of the MU loop (HIP events), best of 3.
usage: r5_geo_scan.py R "hws,hS;hws,hS;..." "wws,wS,wres;..." [cohorts...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
R = int(sys.argv[1])
hs = [tuple(int(x) for x in t.split(",")) for t in sys.argv[2].split(";")]
wsl = [tuple(int(x) for x in t.split(",")) for t in sys.argv[3].split(";")]
Cs = [int(c) for c in sys.argv[4:]] or [1, 2]
ctx = N.Context(0)
X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
ctx.set_X(X)
ks = list(range(2, 17)) if R > 0 else [8]
R = abs(R)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
iters = 500

def run(geo, C):
    os.environ["NMFK_EXP_GEO"] = ",".join(str(g) for g in geo)
    os.environ["NMFK_COHORTS"] = str(C)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=20, maxbaditers=10 ** 9)
    best = 1e9
    for rep in range(3):
        ctx.set_profiling(True)
        ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
        p = ctx.get_profile()
        ctx.set_profiling(False)
        best = min(best, p["mu_loop"]["ms"] / p["mu_loop"]["launches"] / iters)
    return best

print(f"k = {ks[0]}:{ks[-1]} x {R} = {len(ks) * R} units, {iters} iterations; GPU ms per iteration of the MU loop", flush=True)
for hg in hs:
    for wg in wsl:
        out = [f"C={C}: {run((hg[0], hg[1], -1, wg[0], wg[1], wg[2]), C):.4f}" for C in Cs]
        print(f"H ws={hg[0]:2d} S={hg[1]:2d} | W ws={wg[0]:2d} S={wg[1]:2d} res={wg[2]:2d}: " + "; ".join(out), flush=True)
