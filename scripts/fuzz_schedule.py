#!/usr/bin/env python3
"""Differential fuzz of the launch schedule and the deferred check: random shapes, rank sets, restart counts and budgets; the
automatic sweep against (a) the plain order of the check block (NMFK_DEFER_OBJ=0), (b) the per-rank packed-VALU launches
(NMFK_HYB=0, NMFK_MFMA_WIDE=0) -- a different kernel family altogether, (c) round 5: the matrix-pipe launch group as two cohorts on two
streams with the W half-step summing the H partials itself (NMFK_COHORTS=2, NMFK_FUSE_RED=1) against one cohort with reduce launches,
both with the launch geometry pinned (NMFK_TARGET_WGS): every unit must keep its BITS.  Reports the worst relative difference of W*H, of the final
objective and of the monitored objective at the checks, and any difference in iteration counts under the reference's stop rule.
usage: fuzz_schedule.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = N.Context(0)
KEYS = ("NMFK_DEFER_OBJ", "NMFK_HYB", "NMFK_MFMA_WIDE", "NMFK_COHORTS", "NMFK_FUSE_RED", "NMFK_TARGET_WGS")
worst = dict(wh_defer=0.0, wh_valu=0.0, obj_defer=0.0, obj_valu=0.0, trace_defer=0.0)
bad = 0
def rel(a, b, X):
    return float(np.linalg.norm(a - b) / np.linalg.norm(X))
for case in range(ncases):
    n = int(rng.choice([64, 130, 300, 700, 1000, 1500, 2650, 4100]))
    m = int(rng.choice([64, 96, 130, 256, 500, 640, 1000, 2100]))
    nk = int(rng.integers(1, 7))
    ks = sorted(set(int(k) for k in rng.choice(np.arange(2, 41), size=nk, replace=False)))
    R = int(rng.choice([1, 2, 3, 4, 6, 8, 10, 12, 16]))
    maxiter = int(rng.choice([21, 30, 45, 50]))
    planted = bool(rng.integers(0, 2))
    if planted:
        k0 = int(rng.integers(2, 6))
        X = (ctx.fill_uniform(100 + case, 0, n * k0).reshape(k0, n).T.astype(np.float64) @ ctx.fill_uniform(200 + case, 0, k0 * m).reshape(m, k0).T.astype(np.float64)
             + 0.02 * ctx.fill_uniform(300 + case, 0, n * m).reshape(m, n).T).astype(np.float32)
        kw = dict(maxiter=600)  # the reference's stop rule
    else:
        X = (0.05 + ctx.fill_uniform(100 + case, 0, n * m)).reshape(m, n).T.astype(np.float32)
        kw = dict(maxiter=maxiter, maxbaditers=10 ** 9)
    X = np.asfortranarray(X)
    ctx.set_X(X)
    seeds = np.array([[N.run_seed(case + 1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    out, tr = {}, {}
    ctx.set_objective_trace(True)
    for mode, env in (("auto", {}), ("plain", {"NMFK_DEFER_OBJ": "0"}), ("valu", {"NMFK_HYB": "0", "NMFK_MFMA_WIDE": "0"}),
                      ("coh1", {"NMFK_TARGET_WGS": "512", "NMFK_COHORTS": "1", "NMFK_FUSE_RED": "0"}),
                      ("coh", {"NMFK_TARGET_WGS": "512", "NMFK_COHORTS": "2", "NMFK_FUSE_RED": "1"})):
        for key in KEYS:
            os.environ.pop(key, None)
        os.environ.update(env)
        out[mode] = ctx.mu_sweep(ks, R, seeds=seeds, **kw)
        tr[mode] = {(k, r): ctx.objective_trace(ks.index(k), r) for k in ks for r in range(min(R, 2))}
        if mode == "auto":
            info = ctx.last_sweep_info()
        if mode == "coh":
            info_c = ctx.last_sweep_info()
    ctx.set_objective_trace(False)
    line = f"case {case:3d}: {n:5d} x {m:5d} {'planted' if planted else 'noise  '} k = {ks} x {R}, {kw.get('maxiter')} iterations; group units {info['mfma_group_units']}, wide {info['wide_mfma_units']}, groups {info['launch_groups']}, phases {info['phases']}, deferred {info['deferred_checks']} plain {info['plain_checks']}"
    line += f"; forced: cohorts {info_c['cohorts']}, fused reductions {info_c['fused_reductions']}"
    problems = []
    for k in ks:
        if not (np.array_equal(out["coh1"][k]["W"], out["coh"][k]["W"]) and np.array_equal(out["coh1"][k]["H"], out["coh"][k]["H"])
                and np.array_equal(out["coh1"][k]["iters"], out["coh"][k]["iters"])):
            problems.append(f"k={k}: two cohorts + fused reduce changed the unit's bits")
        for other, tag, tol in (("plain", "defer", 2e-5), ("valu", "valu", 2e-4)):
            same = out["auto"][k]["iters"] == out[other][k]["iters"]
            if not planted and not same.all():
                problems.append(f"k={k}: iteration counts differ from {other} under a fixed budget")
            if planted and same.mean() < 0.7:
                problems.append(f"k={k}: only {same.mean():.2f} of the iteration counts equal {other}'s")
            for r in np.flatnonzero(same)[:2]:
                e = rel(out["auto"][k]["W"][r] @ out["auto"][k]["H"][r], out[other][k]["W"][r] @ out[other][k]["H"][r], X)
                worst["wh_" + tag] = max(worst["wh_" + tag], e)
                eo = abs(float(out["auto"][k]["objvalue"][r]) - float(out[other][k]["objvalue"][r])) / max(float(out[other][k]["objvalue"][r]), 1e-30)
                worst["obj_" + tag] = max(worst["obj_" + tag], eo)
                if e > tol or eo > tol:
                    problems.append(f"k={k} r={r}: W*H differs from {other} by {e:.2e}, objective by {eo:.2e}")
        for r in range(min(R, 2)):
            a, b = tr["auto"][(k, r)], tr["plain"][(k, r)]
            nc = min(len(a), len(b))
            if nc:
                et = float(np.max(np.abs(a[:nc] - b[:nc]) / np.maximum(np.abs(b[:nc]), 1e-30)))
                worst["trace_defer"] = max(worst["trace_defer"], et)
                if et > 5e-5:
                    problems.append(f"k={k} r={r}: monitored objective differs from the plain order by {et:.2e}")
        if not np.isfinite(out["auto"][k]["W"]).all() or not np.isfinite(out["auto"][k]["H"]).all():
            problems.append(f"k={k}: non-finite factors")
    bad += len(problems) > 0
    print(line + ("" if not problems else "   <-- " + "; ".join(problems[:3])), flush=True)
print(f"{ncases} cases, {bad} with problems; worst differences: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
