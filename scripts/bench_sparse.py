#!/usr/bin/env python3
"""BASELINE configs[3]: sparse 0.5%-fill fp32 X 100000 x 4096 (MovieLens-scale), k = 2:32, nruns = 16, fixed budget.
Reports ms per MU iteration and the algorithmic HBM rate (2*nnz*8 B + 4*(n+m)*k*4 B per iteration per unit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import nmfk_jl_amd as N

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 32
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
n, m, fill = 100000, 4096, 0.005
rng = np.random.default_rng(3)
nnz = int(n * m * fill)
rows = rng.integers(0, n, nnz)
cols = rng.integers(0, m, nnz)
vals = rng.uniform(1, 5, nnz).astype(np.float32)
X = sp.csc_matrix((vals, (rows, cols)), shape=(n, m))
X.sum_duplicates()
ctx = N.Context(0)
t = time.perf_counter()
ctx.set_X_sparse(X)
print(f"upload + CSR build: {time.perf_counter() - t:.2f}s, nnz kept {ctx.nnz}")
kmin = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ks = list(range(kmin, kmax + 1))
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
ctx.set_profiling(True)
t = time.perf_counter()
res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
dt = time.perf_counter() - t
bytes_alg = sum((2 * ctx.nnz * 8 + 4 * (n + m) * k * 4) * R for k in ks) * iters
flops = sum(8.0 * ctx.nnz * k * R for k in ks) * iters
prof = ctx.get_profile()
for name, e in sorted(prof.items()):
    print(f"  {name:28s} {e['launches']:6d} x {e['ms'] / max(e['launches'], 1):9.4f} ms")
print(f"{len(ks) * R} units: {1e3 * dt / iters:.2f} ms/iter (incl. checks + D2H of results), "
      f"{bytes_alg / dt / 1e9:.0f} GB/s algorithmic = {bytes_alg / dt / 8e12:.1%} of 8 TB/s, {flops / dt / 1e12:.2f} TFLOP/s; "
      f"obj[k={kmin}]={res[kmin]['objvalue'][0]:.3f} obj[k={kmax}]={res[kmax]['objvalue'][0]:.3f}")
