#!/usr/bin/env python3
"""Structured inputs through the split-operand MFMA kernel: X = 1, W = 1, H row d -> which loop rows get lost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmfk_jl_amd as N
n, m, k, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 16
ctx = N.Context(0)
X = np.asfortranarray(np.ones((n, m), np.float32))
ctx.set_X(X)
seeds = np.array([[N.run_seed(1, k, r) for r in range(R)]], dtype=np.uint64)
# W row i = 2^(i % 8) in every signal; H = 1  =>  H half-step numerator of column j = sum_i q_ij w_i
W0 = np.ones((R, n, k), np.float32) * (1.0 + ((np.arange(n)[None, :, None] * int(os.environ.get('IMUL', '1')) + 0 * np.arange(k)[None, None, :]) % int(os.environ.get('PER', '16'))))
H0 = np.ones((R, k, m), np.float32)
out = {}
for hyb in (0, 1):
    os.environ["NMFK_HYB"] = str(hyb); os.environ["NMFK_HYB_MINK"] = "2"; os.environ["NMFK_HYB_PHASES"] = "0"
    out[hyb] = ctx.mu_sweep([k], R, seeds=seeds, Winit={k: W0}, Hinit={k: H0}, maxiter=1, maxbaditers=10 ** 9)[k]
for f in ("H", "W"):
    a, b = out[0][f][0], out[1][f][0]
    print(f, "ref", a.ravel()[:6], "got", b.ravel()[:6])
    print("  ratio got/ref: min %.6f max %.6f" % ((b / a).min(), (b / a).max()))
    np.set_printoptions(linewidth=250, precision=4, suppress=True)
    if f == "H": print((b / a)[:4, :8])
