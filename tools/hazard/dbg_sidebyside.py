#!/usr/bin/env python3
"""Merged sweep with the matrix-pipe group and the packed-VALU group SIDE BY SIDE (NMFK_MERGE_PHASED=0), repeated:
is every repetition bit-identical to the first?  usage: dbg_sidebyside.py [reps] [R]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("NMFK_MERGE_PHASED", "0")
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, m = (int(os.environ.get("N", 700)), int(os.environ.get("M", 130)))
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx = NMFk.Context(0)
ks = [int(v) for v in os.environ.get("KS", "2,3,5,6,8,13,16,20").split(",")]
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ref, bad, prev, badprev = None, {}, None, {}
for rep in range(reps):
    ctx.set_X(X)
    res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=int(os.environ.get("ITERS", 20)), maxbaditers=10 ** 9,
                       **({"compute": NMFk.COMPUTE_F64} if os.environ.get("COMPUTE") == "f64" else {}))
    if ref is None:
        ref = res; print(ctx.last_sweep_info()); continue
    for k in ks:
        if not ((res[k]["W"] == ref[k]["W"]).all() and (res[k]["H"] == ref[k]["H"]).all()):
            bad[k] = bad.get(k, 0) + 1
            if rep == 1:
                d = np.abs(res[k]["W"] - ref[k]["W"]) / np.abs(ref[k]["W"]).max()
                print("k", k, "max rel diff vs first", d.max(), "restarts differing", [int(r) for r in range(R) if not (res[k]["W"][r] == ref[k]["W"][r]).all()])
        if prev is not None and not ((res[k]["W"] == prev[k]["W"]).all() and (res[k]["H"] == prev[k]["H"]).all()):
            badprev[k] = badprev.get(k, 0) + 1
    prev = res
print("reps", reps, "mismatches vs first by rank", bad, "vs previous", badprev)
