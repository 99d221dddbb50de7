import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
oracle.build()
ctx = NMFk.Context(0)
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ctx.set_X(X)
ks, R, iters = [2, 3, 5, 6, 8, 13, 16, 20], 4, 40
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
NOSTOP = dict(maxbaditers=10 ** 9)
def rel(a, b): return np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(X)
runs = {}
for name, env in [("hyb0p127", {"NMFK_HYB": "0", "NMFK_POISON": "127"}), ("hyb0m1p127", {"NMFK_HYB": "0", "NMFK_MERGE": "1", "NMFK_POISON": "127"}), ("defp127", {"NMFK_POISON": "127"}), ("defp1", {"NMFK_POISON": "1"})]:
    os.environ.update(env)
    runs[name] = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, **NOSTOP)
    print(name, ctx.last_sweep_info())
    for k_ in env: del os.environ[k_]
for q, k in enumerate(ks):
    for r in range(R):
        W0, H0 = oracle.init_factors(int(seeds[q, r]), n, m, k)
        ref = oracle.singlerun(X, k, W0, H0, maxiter=iters, **NOSTOP)
        R_ = ref["W"] @ ref["H"]
        vals = [rel(v[k]["W"][r] @ v[k]["H"][r], R_) for v in runs.values()]
        if all(x < 2e-7 for x in vals): continue
        print(k, r, " ".join("%s=%.2e" % (nm, rel(v[k]["W"][r] @ v[k]["H"][r], R_)) for nm, v in runs.items()))
