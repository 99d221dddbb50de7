#!/usr/bin/env python3
"""Soak test of the SHIPPED schedules: the same sweep repeated, every repetition must reproduce the first bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nmfk_jl_amd as NMFk
ctx = NMFk.Context(0)
def case(name, n, m, ks, R, iters, reps, env=None):
    for k_, v_ in (env or {}).items(): os.environ[k_] = v_
    X = np.asfortranarray(0.05 + ctx.fill_uniform(33, 0, n * m).reshape(m, n).T)
    seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref, bad, t0 = None, 0, time.time()
    for rep in range(reps):
        ctx.set_X(X)
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
        if ref is None:
            ref, info = res, ctx.last_sweep_info(); continue
        bad += any(not ((res[k]["W"] == ref[k]["W"]).all() and (res[k]["H"] == ref[k]["H"]).all() and (res[k]["objvalue"] == ref[k]["objvalue"]).all()) for k in ks)
    for k_ in (env or {}): del os.environ[k_]
    print(f"{name:34s} {n}x{m} R={R} iters={iters}: {bad} of {reps - 1} repetitions differ  ({time.time() - t0:.0f} s)  {info}", flush=True)
ks = [2, 3, 5, 6, 8, 13, 16, 20]
case("8 restarts: MFMA group + merged VALU", 700, 130, ks, 8, 40, 200)
case("4 restarts: all on the MFMA group", 700, 130, ks, 4, 40, 200)
case("8 restarts: MFMA group + merged VALU", 8192, 512, list(range(2, 17)), 8, 30, 40)
case("4 restarts: all on the MFMA group", 8192, 512, list(range(2, 17)), 4, 30, 40)
case("two-phase sweep", 2048, 512, list(range(2, 17)), 16, 30, 40)
case("two-phase sweep (bench shape)", 8192, 512, list(range(2, 17)), 32, 20, 12)
case("8 restarts, 400 repetitions", 700, 130, ks, 8, 40, 400)
case("merged kernel on request, side by side", 700, 130, ks, 8, 40, 100, {"NMFK_HYB": "1", "NMFK_HYB_MINK": "6", "NMFK_MERGE": "1"})
case("merged kernel for ALL ranks <= 16", 700, 130, ks, 8, 40, 100, {"NMFK_HYB": "0", "NMFK_MERGE": "2"})
case("packed-VALU only", 700, 130, ks, 8, 40, 100, {"NMFK_HYB": "0"})


def retiring_case(name, n, m, k0, scale, ks, R, reps, env=None):
    """Round 4: the default stop rule on a planted matrix whose restarts retire at different iterations -- the retire-aware
    schedule re-plans the sweep on the way (how often: `replans` in the printed schedule); iteration counts and stop reasons are
    part of what must reproduce."""
    for k_, v_ in (env or {}).items(): os.environ[k_] = v_
    W0 = ctx.fill_uniform(2, 0, n * k0).reshape(k0, n).T.astype(np.float64)
    H0 = ctx.fill_uniform(2, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
    U = ctx.fill_uniform(2, n * k0 + k0 * m, n * m).reshape(m, n).T.astype(np.float64)
    X = np.asfortranarray((scale * (W0 @ H0 + 0.01 * U)).astype(np.float32))
    seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref, bad, t0 = None, 0, time.time()
    for rep in range(reps):
        ctx.set_X(X)
        res = ctx.mu_sweep(ks, R, seeds=seeds)
        if ref is None:
            ref, info = res, ctx.last_sweep_info(); continue
        bad += any(not all((res[k][key] == ref[k][key]).all() for key in ("W", "H", "objvalue", "iters", "reason")) for k in ks)
    for k_ in (env or {}): del os.environ[k_]
    its = np.concatenate([ref[k]["iters"] for k in ks])
    print(f"{name:34s} {n}x{m} R={R} iterations {its.min()}..{its.max()}: {bad} of {reps - 1} repetitions differ  ({time.time() - t0:.0f} s)  {info}", flush=True)


retiring_case("retire-aware schedule (oracle fixture matrix)", 1024, 256, 5, 1.0, list(range(2, 14)), 16, 12, {"NMFK_HYB": "1", "NMFK_HYB_MINK": "2", "NMFK_HYB_PHASES": "1"})
retiring_case("retire-aware, every tier", 640, 192, 3, 1.0, [2, 3, 4, 5, 6], 6, 40, {"NMFK_REPLAN": "2"})
retiring_case("retire-aware (bench shape, 0.03 x)", 8192, 512, 6, 0.03, list(range(2, 17)), 32, 4)


def sparse_case(name, n, m, fill, ks, R, iters, reps, env=None):
    import scipy.sparse as sp
    for k_, v_ in (env or {}).items(): os.environ[k_] = v_
    rng = np.random.default_rng(5)
    nnz = int(n * m * fill)
    Xs = sp.csc_matrix((rng.uniform(1, 5, nnz).astype(np.float32), (rng.integers(0, n, nnz), rng.integers(0, m, nnz))), shape=(n, m))
    Xs.sum_duplicates()
    seeds = np.array([[NMFk.run_seed(12, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref, bad, t0 = None, 0, time.time()
    for rep in range(reps):
        ctx.set_X_sparse(Xs)
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
        if ref is None:
            ref = res; continue
        bad += any(not ((res[k]["W"] == ref[k]["W"]).all() and (res[k]["H"] == ref[k]["H"]).all() and (res[k]["objvalue"] == ref[k]["objvalue"]).all()) for k in ks)
    for k_ in (env or {}): del os.environ[k_]
    print(f"{name:34s} {n}x{m} fill={fill} R={R} iters={iters}: {bad} of {reps - 1} repetitions differ  ({time.time() - t0:.0f} s)", flush=True)


sparse_case("sparse X, blocked form", 20000, 3000, 0.005, [3, 9, 20, 32, 40], 8, 20, 40)
sparse_case("sparse X, blocked form forced", 3000, 1500, 0.01, [3, 9, 20, 32], 4, 20, 60, {"NMFK_SP_BLK": "2"})
sparse_case("sparse X, gather form", 3000, 1500, 0.01, [3, 9, 20, 32, 40], 4, 20, 40, {"NMFK_SP_BLK": "0"})
