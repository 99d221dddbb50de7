#!/usr/bin/env python3
"""Round 5: cohorts of the matrix-pipe launch group (NMFK_COHORTS) and the W half-step's form at very few units, fixed budget at
8192 x 512.  usage: r5_cohorts.py [cohorts|wform|both]   Prints ms per MU iteration; every variant's factors are compared bit for
bit with the first variant of its row (cohorts and the resident / streaming choice... the latter only to 1e-5: another geometry)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
ctx = N.Context(0)
X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
ctx.set_X(X)
what = sys.argv[1] if len(sys.argv) > 1 else "both"

def run(ks, R, iters, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
        ctx.mu_sweep(ks, R, seeds=seeds, maxiter=20, maxbaditers=10 ** 9)
        best = 1e9
        for rep in range(2):
            t = time.perf_counter()
            res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
            best = min(best, time.perf_counter() - t)
        return 1e3 * best / iters, res, ctx.last_sweep_info()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

def same(a, b, ks):
    return all(np.array_equal(a[k]["W"], b[k]["W"]) and np.array_equal(a[k]["H"], b[k]["H"]) for k in ks)

if what in ("cohorts", "both"):
    ks = list(range(2, 17))
    for R, iters in ((32, 300), (16, 400), (8, 600), (4, 1000), (2, 1000), (1, 1000)):
        line, ref = [], None
        for C in (1, 2, 3, 4):
            ms, res, info = run(ks, R, iters, {"NMFK_COHORTS": C})
            ref = ref or res
            line.append(f"C={C}: {ms:.4f} ms{'' if same(ref, res, ks) else ' BITS DIFFER'} (cohorts {info['cohorts']})")
        print(f"k=2:16 x {R:2d} ({15 * R:3d} units): " + "; ".join(line), flush=True)
if what in ("wform", "both"):
    for ks, R in (([8], 1), ([16], 1), ([3], 1), ([8], 2), ([8], 4), ([8], 8), (list(range(2, 17)), 1), (list(range(2, 17)), 2)):
        line, ref = [], None
        for tag, env in (("res2", {"NMFK_HYB_RES": 1, "NMFK_EXP_RES_PAIRS": 2}), ("res1", {"NMFK_HYB_RES": 1, "NMFK_EXP_RES_PAIRS": 1}),
                         ("stream", {"NMFK_HYB_RES": 0})):
            for C in (1, 2):
                if C > 1 and len(ks) * R < 2:
                    continue
                e = dict(env)
                e["NMFK_COHORTS"] = C
                ms, res, info = run(ks, R, 1000, e)
                ref = ref or res
                d = max(float(np.max(np.abs(res[k]["W"] - ref[k]["W"]) / (np.abs(ref[k]["W"]) + 1e-30))) for k in ks)
                line.append(f"{tag} C={C}: {ms:.4f}" + (f" (rel {d:.1e})" if d > 0 else ""))
        print(f"k={ks[0]}..{ks[-1]} x {R} ({len(ks) * R} units): " + "; ".join(line), flush=True)
