import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nmfk_jl_amd as N, nmfk_oracle as oracle
ctx = N.Context(0)
n, m = 333, 275
X = (0.05 + oracle.uniform_fill(61, 0, n * m)).reshape(n, m).astype(np.float32)
ctx.set_X(X)
ks, R = [4, 9, 10, 11, 12, 13, 14, 15, 16, 20], 32
seeds = np.array([[N.run_seed(3, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
NOSTOP = dict(maxbaditers=10 ** 9)
a = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=25, **NOSTOP)
os.environ["NMFK_HYB"] = "0"
b = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=25, **NOSTOP)
del os.environ["NMFK_HYB"]
nx = np.linalg.norm(X)
for k in ks:
    e = max(np.linalg.norm(a[k]["W"][r] @ a[k]["H"][r] - b[k]["W"][r] @ b[k]["H"][r]) / nx for r in range(R))
    print("k", k, f"{e:.2e}")
    assert e < 5e-6
# fixed H with phases
k = 12
W0, H0 = oracle.init_factors(77, n, m, k)
Wi = {kk: np.broadcast_to(oracle.init_factors(77, n, m, kk)[0].astype(np.float32), (R, n, kk)).copy() for kk in ks}
Hi = {kk: np.broadcast_to(oracle.init_factors(77, n, m, kk)[1].astype(np.float32), (R, kk, m)).copy() for kk in ks}
for fixed in ("Hfixed", "Wfixed"):
    res = ctx.mu_sweep(ks, R, Winit=Wi, Hinit=Hi, maxiter=25, normalize=0, **{fixed: 1}, **NOSTOP)
    ref = oracle.singlerun(X, k, W0, H0, maxiter=25, modifymatrices=False, **{fixed: True}, **NOSTOP)
    e = np.linalg.norm(res[k]["W"][5] @ res[k]["H"][5] - ref["W"] @ ref["H"]) / nx
    print(fixed, f"{e:.2e}")
    assert e < 1e-4
print("ok")
