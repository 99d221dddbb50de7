import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nmfk_jl_amd as N, nmfk_oracle as oracle
ctx = N.Context(0)
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx.set_X(X)
ks, R = [2, 3, 5, 8, 13, 16, 20], 4
seeds = np.array([[N.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
a = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=40)
os.environ["NMFK_HYB"] = "1"
for hg in ("1", "2"):
    os.environ["NMFK_HYB_GROUPS"] = hg
    b = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=40)
    for k in ks:
        e = max(np.linalg.norm(a[k]["W"][r] @ a[k]["H"][r] - b[k]["W"][r] @ b[k]["H"][r]) / np.linalg.norm(X) for r in range(R))
        print(hg, k, f"{e:.2e}", np.array_equal(a[k]["iters"], b[k]["iters"]))
        assert e < 5e-6
print("merged hybrid ok")
