#!/usr/bin/env python3
"""Fixed-budget timing of the MU kernels on the BASELINE shape (for A/B of kernel variants on the GPU box).
usage: NMFK_HIP_LIB=path python scripts/microbench.py [iters] [kmin] [kmax] [nruns]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nmfk_jl_amd as N
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kmin = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 16
R = int(sys.argv[4]) if len(sys.argv) > 4 else 32
n, m = 8192, 512
ctx = N.Context(0)
X = ctx.fill_uniform(1, 0, n * m).reshape(m, n).T
ctx.set_X(X)
ks = list(range(kmin, kmax + 1))
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
ctx.set_profiling(True)
t = time.perf_counter()
res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
dt = time.perf_counter() - t
p = ctx.get_profile()
flops = sum(4.0 * n * m * k * R * iters for k in ks)  # per half-step, all launches
line = " ".join(f"{k.replace('mu_', '')}={v['ms'] / max(v['launches'], 1):.3f}ms" for k, v in p.items() if v["launches"])
print(f"{os.environ.get('NMFK_HIP_LIB', 'default'):24s} wall/iter={1e3 * dt / iters:.3f}ms eff={2 * flops / dt / 1e12:.1f}TF | {line} | obj[k={ks[-1]}]={res[ks[-1]]['objvalue'][0]:.6f}")
