#!/usr/bin/env python3
"""Differential fuzz of the sparse path: random sparse X, rank sets, restarts, budgets; the blocked form with the deferred check
(NMFK_SP_BLK=2) against (a) the plain order of the check block (NMFK_DEFER_OBJ=0), (b) the gather form (NMFK_SP_BLK=0).
usage: fuzz_sparse.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import nmfk_jl_amd as N
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = N.Context(0)
worst = dict(wh_defer=0.0, wh_gather=0.0, trace_defer=0.0, trace_gather=0.0)
bad = 0
for case in range(ncases):
    n = int(rng.choice([300, 1100, 2300, 5000, 20000]))
    m = int(rng.choice([96, 300, 1100, 2500]))
    fill = float(rng.choice([0.004, 0.01, 0.03]))
    nnz = max(int(n * m * fill), n + m)
    Xs = sp.csc_matrix((rng.uniform(1, 5, nnz).astype(np.float32), (rng.integers(0, n, nnz), rng.integers(0, m, nnz))), shape=(n, m))
    Xs = Xs + sp.csc_matrix((np.full(n, 0.5, np.float32), (np.arange(n), np.arange(n) % m)), shape=(n, m)) \
            + sp.csc_matrix((np.full(m, 0.5, np.float32), (np.arange(m) % n, np.arange(m))), shape=(n, m))  # no empty row / column
    Xs = sp.csc_matrix(Xs); Xs.sum_duplicates()
    nk = int(rng.integers(1, 6))
    ks = sorted(set(int(k) for k in rng.choice(np.arange(2, 41), size=nk, replace=False)))
    R = int(rng.choice([1, 2, 4, 8]))
    maxiter = int(rng.choice([21, 30, 45]))
    seeds = np.array([[N.run_seed(case + 1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    out, tr = {}, {}
    ctx.set_objective_trace(True)
    for mode, env in (("auto", {"NMFK_SP_BLK": "2"}), ("plain", {"NMFK_SP_BLK": "2", "NMFK_DEFER_OBJ": "0"}), ("gather", {"NMFK_SP_BLK": "0"})):
        for key in ("NMFK_SP_BLK", "NMFK_DEFER_OBJ"):
            os.environ.pop(key, None)
        os.environ.update(env)
        ctx.set_X_sparse(Xs)
        out[mode] = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=maxiter, maxbaditers=10 ** 9)
        tr[mode] = {(k, r): ctx.objective_trace(ks.index(k), r) for k in ks for r in range(min(R, 2))}
        if mode == "auto":
            info = ctx.last_sweep_info()
    ctx.set_objective_trace(False)
    Xd_norm = float(np.sqrt(Xs.multiply(Xs).sum()))
    problems = []
    for k in ks:
        for other, tag, tol in (("plain", "defer", 2e-5), ("gather", "gather", 2e-4)):
            if not (out["auto"][k]["iters"] == out[other][k]["iters"]).all():
                problems.append(f"k={k}: iteration counts differ from {other}")
            for r in range(min(R, 2)):
                e = float(np.linalg.norm(out["auto"][k]["W"][r] @ out["auto"][k]["H"][r] - out[other][k]["W"][r] @ out[other][k]["H"][r]) / Xd_norm)
                worst["wh_" + tag] = max(worst["wh_" + tag], e)
                if e > tol:
                    problems.append(f"k={k} r={r}: W*H differs from {other} by {e:.2e}")
                a, b = tr["auto"][(k, r)], tr[other][(k, r)]
                nc = min(len(a), len(b))
                if nc:
                    et = float(np.max(np.abs(a[:nc] - b[:nc]) / np.maximum(np.abs(b[:nc]), 1e-30)))
                    worst["trace_" + tag] = max(worst["trace_" + tag], et)
                    if et > 1e-4:
                        problems.append(f"k={k} r={r}: monitored objective differs from {other} by {et:.2e}")
    bad += len(problems) > 0
    print(f"case {case:3d}: {n:5d} x {m:4d} fill {fill} k = {ks} x {R}, {maxiter} iterations; deferred {info['deferred_checks']} plain {info['plain_checks']}, groups {info['launch_groups']}"
          + ("" if not problems else "   <-- " + "; ".join(problems[:3])), flush=True)
print(f"{ncases} cases, {bad} with problems; worst differences: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
