#!/usr/bin/env python3
"""Per-rank GPU time of the bench sweep when sharded over N ranks by parallel.plan_shards: every rank's share is run
alone on this GPU, one after the other; the slowest share is what strong scaling can reach before communication.
usage: rank_sim.py [--matrix noise|planted|retiring] [N ...]
  noise     U(0,1) (the bench matrix: every restart runs to maxiter, the static shards are equal by construction)
  planted   SURVEY 8d's rank-6 matrix W0 H0 + 0.01 U (restarts stop between 1 000 and 10 000 iterations, 91 % of the unit-slots live)
  retiring  0.1 * planted: the stop rule's absolute tolOF retires restarts between 1 000 and 10 000 iterations (44 % live)
For the structured matrices the line also says how unequal the static shards {g, g + N, ...} are (work = sum over a rank's
restarts of k * iterations) -- what the reference's dynamic hand-out (Distributed.pmap, src/NMFkExecute.jl:516-518) would even out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m, R = 8192, 512, 32
ks = list(range(2, 17))
ctx = N.Context(0)
args = sys.argv[1:]
matrix = "noise"
if args and args[0] == "--matrix":
    matrix, args = args[1], args[2:]
if matrix == "noise":
    X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
else:
    k0 = 6
    W0 = ctx.fill_uniform(2, 0, n * k0).reshape(k0, n).T.astype(np.float64)
    H0 = ctx.fill_uniform(2, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
    U = ctx.fill_uniform(2, n * k0 + k0 * m, n * m).reshape(m, n).T.astype(np.float64)
    X = np.asfortranarray(((0.1 if matrix == "retiring" else 1.0) * (W0 @ H0 + 0.01 * U)).astype(np.float32))
ctx.set_X(X)
print(f"matrix: {matrix}", flush=True)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, 2, seeds=seeds[:, :2], maxiter=20)
base = None
for nr in [int(a) for a in args] or [1, 2, 4, 8]:
    c, chunks = N.parallel.plan_shards(ks, R, nr)  # [(kidx, restarts, owner)]
    times, work = [], []
    for g in range(1 if os.environ.get("RANK_SIM_FIRST") else nr):  # RANK_SIM_FIRST=1: rank 0's share only
        mine = [ch for ch in chunks if ch[2] == g]
        lks = [ks[q] for q, *_ in mine]
        sd = np.stack([seeds[q, rs + [rs[-1]] * (c - len(rs))] for q, rs, _ in mine])
        t = time.perf_counter()
        res = ctx.mu_sweep(lks, c, seeds=sd)
        times.append(time.perf_counter() - t)
        work.append(sum(float(k) * float(np.sum(res[k]["iters"][:len(rs)])) for k, (q, rs, _) in zip(lks, mine)))
    base = base or max(times)
    print(f"N={nr}: {c} restarts x {len(mine)} ranks per GPU; per-rank seconds {' '.join(f'{t:.2f}' for t in times)}; "
          f"slowest {max(times):.2f} s, ideal {base / nr:.2f} s, efficiency {base / nr / max(times):.2f}; "
          f"static shards: work (sum k * iterations) max / mean = {max(work) / (sum(work) / len(work)):.3f}, "
          f"seconds max / mean = {max(times) / (sum(times) / len(times)):.3f}", flush=True)
