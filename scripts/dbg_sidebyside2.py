#!/usr/bin/env python3
"""Where do the merged packed-VALU units differ run to run beside the MFMA group (NMFK_MERGE_PHASED=0)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("NMFK_MERGE_PHASED", "0")
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
iters = int(sys.argv[1]); R = 8
n, m = int(os.environ.get("N", 700)), int(os.environ.get("M", 130))
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx = NMFk.Context(0)
ks = [2, 3, 5, 6, 8, 13, 16, 20]
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
os.environ["NMFK_STREAMS"] = "1"
ctx.set_X(X); ref = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
del os.environ["NMFK_STREAMS"]
for rep in range(int(os.environ.get('REPS', 6))):
    ctx.set_X(X); res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
    for k in (2, 3, 5):
        for f in ("H", "W"):
            a, b = ref[k][f], res[k][f]
            bad = np.argwhere(a != b)
            if len(bad):
                rs = sorted(set(bad[:, 0].tolist()))
                r0 = rs[0]
                sub = bad[bad[:, 0] == r0]
                print(f"rep {rep} k {k} {f}: restarts {rs}; restart {r0}: {len(sub)} entries differ, rows {sorted(set(sub[:,1].tolist()))[:12]} cols {sorted(set(sub[:,2].tolist()))[:12]} maxrel {np.abs(a-b).max()/np.abs(a).max():.2e}")
