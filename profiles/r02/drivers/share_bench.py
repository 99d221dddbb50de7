#!/usr/bin/env python3
"""One rank's share of the bench sweep at N GPUs (R = 32 / N restarts of every rank k = 2:16), fixed budget: GPU time of
the MU loop and per-kernel launch averages.  usage: share_bench.py N [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
nr = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n, m, R = 8192, 512, 32 // nr
ks = list(range(2, 17))
ctx = N.Context(0)
ctx.set_X(ctx.fill_uniform(1, 0, n * m).reshape(m, n).T)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
ctx.set_profiling(True)
ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
p = ctx.get_profile()
info = ctx.last_sweep_info()
line = " ".join(f"{k}={v['ms'] / max(v['launches'], 1):.3f}" for k, v in p.items() if v["launches"] and k != "mu_loop")
print(f"N={nr} {os.environ.get('TAG', ''):34s} loop {p['mu_loop']['ms']:8.2f} ms / {iters} it  phases={info['phases']} mfma_units={info['mfma_group_units']} | {line}")
