#!/usr/bin/env python3
"""One MU iteration on the split-operand MFMA kernel vs the packed-VALU kernel: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m, k, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 16
ctx = N.Context(0)
X = np.asfortranarray(0.05 + ctx.fill_uniform(7, 0, n * m).reshape(m, n).T)
ctx.set_X(X)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)]], dtype=np.uint64)
out = {}
for hyb in (0, 1):
    os.environ["NMFK_HYB"] = str(hyb); os.environ["NMFK_HYB_MINK"] = "2"; os.environ["NMFK_HYB_PHASES"] = "0"
    out[hyb] = ctx.mu_sweep([k], R, seeds=seeds, maxiter=int(os.environ.get("ITERS", "1")), maxbaditers=10 ** 9)[k]
    print(hyb, ctx.last_sweep_info())
for f in ("W", "H"):
    a, b = out[0][f][0], out[1][f][0]
    d = np.abs(a - b) / (np.abs(a) + 1e-30)
    print(f, a.shape, "nan", np.isnan(b).sum(), "max rel diff", np.nanmax(d), "at", np.unravel_index(np.nanargmax(d), d.shape))
    bad = np.argwhere(~(d < 1e-4))
    print(" bad count", len(bad), "first", bad[:8].tolist())
    if len(bad):
        i, j = bad[0]
        print(" ref", a[i, j], "got", b[i, j])
        print(" bad rows", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist()))[:20])
np.set_printoptions(linewidth=250, precision=3, suppress=True)
a, b = out[0]["H"][0], out[1]["H"][0]
r = (b / a - 1) * 100
print("H rel err % by signal (rows) x first 16 columns"); print(r[:, :16])
print("mean |err| per signal", np.abs(r).mean(axis=1))
print("mean |err| per column mod 32", np.array([np.abs(r[:, j::32]).mean() for j in range(32)]))
print("mean err (signed) per signal", r.mean(axis=1))
