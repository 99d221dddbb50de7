#!/usr/bin/env python3
"""The SHIPPED schedules beside a neighbour that keeps gfx950's 128-bit-operand matrix instructions busy on every CU
(tools/hazard/burner.hip mode 0, another process; tools/hazard/soak_beside_burner.sh).  Every reference is computed first, with the
GPU to ourselves; then the burner starts and every repetition must reproduce its reference bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import nmfk_jl_amd as NMFk
ctx = NMFk.Context(0)
scale = float(os.environ.get("REPS_SCALE", "1"))

def same(a, b):
    if isinstance(a, dict):
        return all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is b
    if isinstance(a, str):
        return a == b
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()

cases = []
def sweep(name, n, m, ks, R, iters, reps, **kw):
    X = np.asfortranarray(0.05 + ctx.fill_uniform(33, 0, n * m).reshape(m, n).T)
    seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    def run():
        ctx.set_X(X)
        res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9, **kw)
        return {k: {f: res[k][f] for f in ("W", "H", "objvalue")} for k in ks}
    cases.append((f"{name}  {n}x{m} R={R} iters={iters}", run, reps))

ks = [2, 3, 5, 6, 8, 13, 16, 20]
sweep("8 restarts: MFMA group + merged VALU", 700, 130, ks, 8, 40, 150)
sweep("4 restarts: all on the MFMA group", 700, 130, ks, 4, 40, 150)
sweep("32 restarts: two phases", 2048, 512, list(range(2, 17)), 32, 20, 20)
sweep("fp64 compute (merged fp64 kernel)", 700, 130, [2, 3, 5, 8], 4, 20, 60, compute=NMFk.COMPUTE_F64)
sweep("8 restarts, bench shape: MFMA group + merged VALU", 8192, 512, list(range(2, 17)), 8, 20, 12)
sweep("32 restarts, bench shape: two phases", 8192, 512, list(range(2, 17)), 32, 10, 6)
sweep("wide ranks", 1024, 256, [20, 32, 48, 64], 4, 20, 40)
sweep("6 restarts, ranks 2:12 (merged VALU beside the group)", 1500, 300, list(range(2, 13)), 6, 30, 40)

def sparse_case():
    n, m = 6000, 700
    rng = np.random.default_rng(5)
    nnz = int(n * m * 0.01)
    X = sp.csc_matrix((rng.uniform(1, 5, nnz).astype(np.float32), (rng.integers(0, n, nnz), rng.integers(0, m, nnz))), shape=(n, m))
    X.sum_duplicates()
    kss = [3, 8, 13, 20, 32]
    seeds = np.array([[NMFk.run_seed(7, k, r) for r in range(4)] for k in kss], dtype=np.uint64)
    def run():
        ctx.set_X_sparse(X)
        res = ctx.mu_sweep(kss, 4, seeds=seeds, maxiter=20, maxbaditers=10 ** 9)
        return {k: {f: res[k][f] for f in ("W", "H", "objvalue")} for k in kss}
    cases.append((f"sparse gather kernels  {n}x{m} 1 % fill R=4 iters=20", run, 40))
sparse_case()

def execute_case():
    n, m = 600, 96
    W0 = ctx.fill_uniform(3, 0, n * 4).reshape(n, 4); H0 = ctx.fill_uniform(4, 0, 4 * m).reshape(4, m)
    X = np.asfortranarray((W0 @ H0 + 0.01 * ctx.fill_uniform(5, 0, n * m).reshape(n, m)).astype(np.float32))
    def run():
        out = NMFk.execute(X, range(2, 8), 8, load=False, save=False, quiet=True, seed=3, ctx=ctx, maxiter=300)
        return [list(out[0]), list(out[1]), out[2], out[3], out[4], out[5]]
    cases.append((f"execute: sweep, clustering, silhouettes, kopt  {n}x{m} k=2:7 R=8", run, 40))
execute_case()

def retiring_case():
    """Round 4: the default stop rule on a planted matrix (restarts stop at different iterations), every tier of the
    retire-aware schedule taken (NMFK_REPLAN=2): replan_kernel, the one-walk clamp and the lowflag of the fused finishes beside the burner."""
    n, m, k0 = 640, 192, 3
    W0 = ctx.fill_uniform(2, 0, n * k0).reshape(k0, n).T.astype(np.float64); H0 = ctx.fill_uniform(2, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
    X = np.asfortranarray((W0 @ H0 + 0.01 * ctx.fill_uniform(2, n * k0 + k0 * m, n * m).reshape(m, n).T).astype(np.float32))
    kss = [2, 3, 4, 5, 6]
    seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(6)] for k in kss], dtype=np.uint64)
    def run():
        os.environ["NMFK_REPLAN"] = "2"
        try:
            ctx.set_X(X)
            res = ctx.mu_sweep(kss, 6, seeds=seeds, maxiter=3000)
            assert ctx.last_sweep_info()["replans"] >= 2
        finally:
            del os.environ["NMFK_REPLAN"]
        return {k: {f: res[k][f] for f in ("W", "H", "objvalue", "iters", "reason")} for k in kss}
    cases.append((f"retire-aware schedule, every tier  {n}x{m} k=2:6 R=6 default stop rule", run, 30))
retiring_case()

def kmeans_case():
    W = np.ascontiguousarray(ctx.fill_uniform(9, 0, 2048 * 6).reshape(2048, 6).astype(np.float32).T)  # 6 x 2048 samples
    def run():
        r = NMFk.robustkmeans(W, 4, 50, seed=11, ctx=ctx)
        return list(r) if isinstance(r, tuple) else r
    cases.append(("robustkmeans  2048 samples of 6, k=4, 50 repeats", run, 40))
kmeans_case()

refs = [run() for _, run, _ in cases]
assert all(same(run(), ref) for (_, run, _), ref in zip(cases[:2], refs[:2])), "not reproducible even alone"
print("references computed alone;", len(cases), "cases", flush=True)
hs = os.environ.get("HANDSHAKE")
if hs:
    open(hs + ".ref", "w").close()
    while not os.path.exists(hs + ".go"):
        time.sleep(0.2)
total = 0
for (name, run, reps), ref in zip(cases, refs):
    reps = max(2, int(reps * scale)); t0 = time.time()
    bad = sum(not same(run(), ref) for _ in range(reps))
    total += bad
    print(f"{name:88s} {bad} of {reps} repetitions differ  ({time.time() - t0:.0f} s)", flush=True)
print("TOTAL differing:", total)
