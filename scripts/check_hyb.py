#!/usr/bin/env python3
"""Split-operand MFMA half-step (nmfk_step_hyb.hip) against the packed-VALU kernel and the CPU oracle on small shapes
(GPU box).  usage: python scripts/check_hyb.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as N
import nmfk_oracle as oracle

NOSTOP = dict(maxbaditers=10 ** 9)
ctx = N.Context(0)
worst = 0.0
for (n, m), k, R, iters in [((64, 32), 5, 6, 60), ((300, 70), 7, 6, 60), ((257, 129), 16, 5, 60), ((130, 2100), 9, 5, 40),
                            ((2100, 96), 13, 5, 40), ((8192, 512), 8, 5, 20), ((8192, 512), 16, 32, 20)]:
    X = oracle.uniform_fill(11, 0, n * m).reshape(n, m).astype(np.float32)
    ctx.set_X(X)
    seeds = np.array([[N.run_seed(5, k, r) for r in range(R)]], dtype=np.uint64)
    os.environ["NMFK_HYB"] = "1"
    a = ctx.mu_sweep([k], R, seeds=seeds, maxiter=iters, **NOSTOP)[k]
    os.environ["NMFK_HYB"] = "0"
    b = ctx.mu_sweep([k], R, seeds=seeds, maxiter=iters, **NOSTOP)[k]
    nx = np.linalg.norm(X)
    for r in range(min(R, 2)):
        W0, H0 = oracle.init_factors(int(seeds[0, r]), n, m, k)
        ref = oracle.singlerun(np.asfortranarray(X), k, W0, H0, maxiter=iters, nthreads=8, **NOSTOP)
        pr = ref["W"] @ ref["H"]
        ea = np.linalg.norm(a["W"][r] @ a["H"][r] - pr) / nx
        eb = np.linalg.norm(b["W"][r] @ b["H"][r] - pr) / nx
        eab = np.linalg.norm(a["W"][r] @ a["H"][r] - b["W"][r] @ b["H"][r]) / nx
        worst = max(worst, ea)
        print(f"{n}x{m} k={k} r={r}: hyb-vs-oracle {ea:.2e}  valu-vs-oracle {eb:.2e}  hyb-vs-valu {eab:.2e}  "
              f"obj {a['objvalue'][r]:.7g} / {b['objvalue'][r]:.7g} / {ref['objvalue']:.7g}", flush=True)
print("worst hyb-vs-oracle", worst)
assert worst < 1e-4
