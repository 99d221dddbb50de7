#!/usr/bin/env python3
"""Does the reference's (fp64) stop rule fire before maxiter on the BASELINE X?  fp64 compute mode reproduces the
oracle's iteration counts (tests/test_gpu_parity.py::test_stop_rule_fp64_identical_iterations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
ctx = N.Context(0)
X = ctx.fill_uniform(1, 0, n * m).reshape(m, n).T
ctx.set_X(X)
ks = [2, 5, 9, 16]
seeds = np.array([[N.run_seed(1, k, r) for r in range(2)] for k in ks], dtype=np.uint64)
for comp in ("f64", "f32"):
    t = time.time()
    res = ctx.mu_sweep(ks, 2, seeds=seeds, compute=N.COMPUTE_F64 if comp == "f64" else N.COMPUTE_F32)
    print(comp, f"{time.time() - t:.1f}s", {k: (res[k]["iters"].tolist(), res[k]["reason"].tolist(), [round(float(v), 4) for v in res[k]["objvalue"]]) for k in ks})
