#!/usr/bin/env python3
"""Round 4: the retire-aware schedule (nmfk_mu_sweep, "tiers") against the static one on structured data.
usage (GPU box): python scripts/r4_replan.py [rank0 noise nruns maxiter scale]
Planted matrix of SURVEY 8d: X = scale * (W0 H0 + noise * U) at 8192 x 512, k = 2:16 x nruns, the reference's default stop rule;
NMFK_REPLAN=0 / 1 in the same process on the same seeds.  Prints seconds per sweep, iteration statistics and how far
the two schedules' results are apart."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nmfk_jl_amd as N

k0 = int(sys.argv[1]) if len(sys.argv) > 1 else 6
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
maxiter = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0  # the stop rule's tolOF = 1e-3 is ABSOLUTE on the sum of squares (Mult:24, 81)
n, m = 8192, 512
ks = list(range(2, 17))
ctx = N.Context(0)
W0 = ctx.fill_uniform(2, 0, n * k0).reshape(k0, n).T.astype(np.float64)
H0 = ctx.fill_uniform(2, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
U = ctx.fill_uniform(2, n * k0 + k0 * m, n * m).reshape(m, n).T.astype(np.float64)
X = np.asfortranarray((scale * (W0 @ H0 + noise * U)).astype(np.float32))
ctx.set_X(X)
seeds = np.array([[N.run_seed(2, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, 2, seeds=seeds[:, :2].copy(), maxiter=20)  # warm-up (tiled X, arena)
out = {}
for mode in (("0", "1") if os.environ.get("R4_QUICK") else ("0", "1", "0", "1")):
    os.environ["NMFK_REPLAN"] = mode
    t = time.perf_counter()
    res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=maxiter)
    dt = time.perf_counter() - t
    it = np.concatenate([res[k]["iters"] for k in ks])
    info = ctx.last_sweep_info()
    print(f"{scale} * (planted rank {k0} + {noise} U), k=2:16 x {R}: NMFK_REPLAN={mode}  {dt:8.3f} s  {len(it) / dt:7.2f} factorizations/s  "
          f"iterations min/mean/max {it.min()}/{it.mean():.0f}/{it.max()}  active unit-slots {it.sum() / (it.max() * len(it)):.3f}  "
          f"info {info}", flush=True)
    out.setdefault(mode, res)
a, b = out["0"], out["1"]
same_it = np.mean(np.concatenate([a[k]["iters"] == b[k]["iters"] for k in ks]))
worst = 0.0
for k in ks:
    for r in range(R):
        if a[k]["iters"][r] == b[k]["iters"][r]:
            Pa = a[k]["W"][r].astype(np.float64) @ a[k]["H"][r].astype(np.float64)
            Pb = b[k]["W"][r].astype(np.float64) @ b[k]["H"][r].astype(np.float64)
            worst = max(worst, np.linalg.norm(Pa - Pb) / np.linalg.norm(X))
            break  # one restart per rank
obj = max(float(np.max(np.abs(a[k]["objvalue"] - b[k]["objvalue"]) / a[k]["objvalue"])) for k in ks)
print(f"static vs retire-aware: iteration counts equal on {100 * same_it:.1f} % of the restarts, worst |WH_a - WH_b| / |X| "
      f"(restarts with equal counts, one per rank) {worst:.2e}, worst relative objective difference {obj:.2e}")
