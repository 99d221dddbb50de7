#!/usr/bin/env python3
"""Round 6 A/B of the wide-rank half-step (wide2_step_kernel): numerators on the bf16 matrix pipe (NMFK_WIDE_BN=1, default) against the
fp32 matrix pipe (NMFK_WIDE_BN=0), same library, same process, alternating.  `python scripts/r6_wide_ab.py [iters] [n] [m] [R]`.
Prints GPU ms per MU iteration (mu_loop of nmfk_get_profile), the algorithmic TFLOP/s (8 n m k per iteration and unit) and the difference
of the two forms' W*H after the budget, relative to ||X||."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nmfk_jl_amd as NMFk

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
m = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
R = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ks = [int(k) for k in os.environ.get("KS", "64,48,40,32,24").split(",")]
ctx = NMFk.Context(0)
X = ctx.fill_uniform(4, 0, n * m).reshape(m, n).T
ctx.set_X(X)
xn = float(np.linalg.norm(X.astype(np.float64)))
rows = np.arange(0, n, max(1, n // 512))
out = []
for k in ks:
    seeds = np.array([[NMFk.run_seed(1, k, r) for r in range(R)]], dtype=np.uint64)
    res, ms = {}, {}
    for rep in range(int(os.environ.get('REPS', '5'))):
        for bn in ("0", "1"):
            os.environ["NMFK_WIDE_BN"] = bn
            ctx.set_profiling(False)
            ctx.mu_sweep([k], R, seeds=seeds, maxiter=2, maxbaditers=10 ** 9)
            ctx.set_profiling(True)
            r = ctx.mu_sweep([k], R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)[k]
            prof = ctx.get_profile()
            ctx.set_profiling(False)
            ms.setdefault(bn, []).append(prof["mu_loop"]["ms"] / iters)
            res[bn] = (r["W"][0][rows].astype(np.float64) @ r["H"][0].astype(np.float64), float(r["objvalue"][0]))
            halves = {nm: round(v["ms"] / v["launches"], 4) for nm, v in prof.items() if nm.startswith(("h_step", "w_step")) and v["launches"]}
            if rep == 1 and os.environ.get('HALVES'):
                print(f"  k={k} bn={bn} half-steps avg ms {halves}", flush=True)
    d = float(np.linalg.norm(res["0"][0] - res["1"][0]) / (xn * np.sqrt(len(rows) / n)))
    tf = lambda v: 8.0 * n * m * k * R / (v * 1e-3) / 1e12
    line = dict(k=k, ms_fp32_numerators=[round(v, 4) for v in ms["0"]], ms_bf16_numerators=[round(v, 4) for v in ms["1"]],
                TFLOPs_fp32=round(tf(min(ms["0"])), 1), TFLOPs_bf16=round(tf(min(ms["1"])), 1), speedup=round(min(ms["0"]) / min(ms["1"]), 3),
                rel_diff_WH=d, objvalue=[res["0"][1], res["1"][1]])
    print(json.dumps(line), flush=True)
    out.append(line)
os.environ.pop("NMFK_WIDE_BN", None)
