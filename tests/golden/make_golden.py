#!/usr/bin/env python3
"""Generates tests/golden/mu_golden.npz from the CPU oracle (run in the build container: `python tests/golden/make_golden.py`).
The fixture holds inputs and expected outputs only (data); the GPU parity tests compare libnmfk_hip against it so that
parity can be checked even where the oracle library is not rebuilt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nmfk_oracle as oracle  # noqa: E402


def main():
    out = {}
    # case A: fixed budget, dense
    n, m, k, R, iters = 48, 20, 4, 3, 40
    X = oracle.uniform_fill(101, 0, n * m).reshape(n, m).astype(np.float32)
    seeds = np.array([oracle.run_seed(9, k, r) for r in range(R)], dtype=np.uint64)
    Ws, Hs, obj = [], [], []
    for r in range(R):
        W0, H0 = oracle.init_factors(int(seeds[r]), n, m, k)
        res = oracle.singlerun(X, k, W0, H0, maxiter=iters, maxbaditers=10 ** 9)
        Ws.append(res["W"]); Hs.append(res["H"]); obj.append(res["objvalue"])
    out.update(A_X=X, A_seeds=seeds, A_k=k, A_iters=iters, A_W=np.stack(Ws), A_H=np.stack(Hs), A_obj=np.array(obj))
    # case B: missing data + a zero, fixed budget
    n, m, k, R, iters = 40, 24, 3, 2, 30
    X = oracle.uniform_fill(102, 0, n * m).reshape(n, m).astype(np.float32)
    mask = oracle.uniform_fill(103, 0, n * m).reshape(n, m) < 0.15
    X[mask] = np.nan
    i0, j0 = np.argwhere(~mask)[3]
    X[i0, j0] = 0
    seeds = np.array([oracle.run_seed(10, k, r) for r in range(R)], dtype=np.uint64)
    Ws, Hs, obj = [], [], []
    for r in range(R):
        W0, H0 = oracle.init_factors(int(seeds[r]), n, m, k)
        res = oracle.singlerun(X, k, W0, H0, maxiter=iters, maxbaditers=10 ** 9)
        Ws.append(res["W"]); Hs.append(res["H"]); obj.append(res["objvalue"])
    out.update(B_X=X, B_seeds=seeds, B_k=k, B_iters=iters, B_W=np.stack(Ws), B_H=np.stack(Hs), B_obj=np.array(obj))
    # case C: default stop rule on the README construction (Readme.md:97-106), whole execute()
    u = oracle.uniform_fill(7, 0, 45).reshape(3, 15)
    a, b, c = u
    X = np.stack([a + 3 * c, 10 * a + b, b, 5 * b + c, a + 2 * b + 5 * c], axis=1).astype(np.float32)
    W, H, fit, rob, aic, kopt, det = oracle.execute(X, range(2, 6), 8, seed=77)
    out.update(C_X=X, C_seed=77, C_nNMF=8, C_fit=fit, C_rob=rob, C_aic=aic, C_kopt=kopt,
               C_iters=np.stack([np.array(det[k]["iters"]) for k in range(2, 6)]))
    # case D: clustering + silhouettes on a fixed stack (T = Float32)
    rng = np.random.default_rng(5)
    base = rng.random((5, 12)) ** 3
    Hs = np.stack([base[rng.permutation(5)] * (1 + 0.05 * rng.random((5, 12))) for _ in range(9)]).astype(np.float32)
    labels, cent = oracle.clustersolutions(list(Hs), tbits=32)
    D, psil, csil = oracle.finalize_silhouettes(list(Hs), labels, tbits=32)
    out.update(D_H=Hs, D_labels=labels, D_centroids=cent, D_psil=psil, D_csil=csil)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mu_golden.npz"), **out)
    print("wrote mu_golden.npz:", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
