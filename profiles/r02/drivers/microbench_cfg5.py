#!/usr/bin/env python3
"""Fixed-budget timing at the BASELINE configs[4] shape (65536 x 2048, k = 64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n, m = 65536, 2048
ctx = N.Context(0)
X = ctx.fill_uniform(4, 0, n * m).reshape(m, n).T
ctx.set_X(X)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)]], dtype=np.uint64)
ctx.mu_sweep([k], R, seeds=seeds, maxiter=2, maxbaditers=10 ** 9)
ctx.set_profiling(True)
t = time.perf_counter()
ctx.mu_sweep([k], R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
dt = time.perf_counter() - t
fl = 8.0 * n * m * k * R * iters
p = ctx.get_profile()
line = " ".join(f"{kk}={v['ms'] / max(v['launches'], 1):.3f}ms({v['flops'] / max(v['ms'], 1e-9) / 1e9:.1f}TF)" for kk, v in p.items() if v["launches"])
print(f"k={k} R={R}: {1e3 * dt / iters:.2f} ms/iter, {fl / dt / 1e12:.1f} TFLOP/s end to end | {line}")
