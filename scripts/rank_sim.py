#!/usr/bin/env python3
"""Per-rank GPU time of the bench sweep when sharded over N ranks by parallel.plan_shards: every rank's share is run
alone on this GPU, one after the other; the slowest share is what strong scaling can reach before communication.
usage: rank_sim.py [N ...]"""
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
base = None
for nr in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    c, chunks = N.parallel.plan_shards(ks, R, nr)  # [(kidx, restarts, owner)]
    times = []
    for g in range(1 if os.environ.get("RANK_SIM_FIRST") else nr):  # RANK_SIM_FIRST=1: rank 0's share only
        mine = [ch for ch in chunks if ch[2] == g]
        lks = [ks[q] for q, *_ in mine]
        sd = np.stack([seeds[q, rs + [rs[-1]] * (c - len(rs))] for q, rs, _ in mine])
        t = time.perf_counter()
        ctx.mu_sweep(lks, c, seeds=sd)
        times.append(time.perf_counter() - t)
    base = base or max(times)
    print(f"N={nr}: {c} restarts x {len(mine)} ranks per GPU; per-rank seconds {' '.join(f'{t:.2f}' for t in times)}; "
          f"slowest {max(times):.2f} s, ideal {base / nr:.2f} s, efficiency {base / nr / max(times):.2f}", flush=True)
