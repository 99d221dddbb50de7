#!/usr/bin/env python3
"""Round 5: is the host (launch calls) or the GPU the bottleneck of few-unit sweeps?  NMFK_HOST_TIMING=1 makes nmfk_mu_sweep print
the loop's host time and how much of it was spent WAITING for the GPU (hipEventSynchronize on the previous check's snapshot): a host
that waits is ahead of the GPU -- the queues are full, there is no launch gap for a hipGraph to remove."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMFK_HOST_TIMING"] = "1"
import numpy as np
import nmfk_jl_amd as N
n, m = 8192, 512
ctx = N.Context(0)
X = ctx.fill_uniform(20260101, 0, n * m).reshape(m, n).T
ctx.set_X(X)
for ks, R in ((list(range(2, 17)), 4), (list(range(2, 17)), 1), ([8], 1)):
    seeds = np.array([[N.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=20, maxbaditers=10 ** 9)
    print(f"k = {ks[0]}:{ks[-1]} x {R} ({len(ks) * R} units), 2000 iterations:", flush=True)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=2000, maxbaditers=10 ** 9)
    sys.stderr.flush()
