import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
ctx = NMFk.Context(0)
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
ks, R, iters = [2, 3, 5, 6, 8, 13, 16, 20], 4, int(os.environ.get("ITERS", "40"))
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
NOSTOP = dict(maxbaditers=10 ** 9)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ref = None
bad = 0
if os.environ.get("KS"): ks = [int(v) for v in os.environ["KS"].split(",")]
seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
perk = {k: 0 for k in ks}
for i in range(reps):
    ctx.set_X(X)
    res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, **NOSTOP)
    if ref is None:
        ref = res
        continue
    for k in ks:
        d = [r for r in range(R) if not ((res[k]["W"][r] == ref[k]["W"][r]).all() and (res[k]["H"][r] == ref[k]["H"][r]).all())]
        if d:
            bad += 1
            perk[k] += 1
            r0 = d[0]
            Wd, Wr, Hd, Hr = res[k]["W"][r0], ref[k]["W"][r0], res[k]["H"][r0], ref[k]["H"][r0]
            print("   W@H rel diff %.2e; col ratio W %s; row ratio H %s; nW diff %d nH diff %d; iters %s vs %s; obj %s vs %s" % (
                np.linalg.norm(Wd @ Hd - Wr @ Hr) / np.linalg.norm(X), np.round(Wd.sum(0) / Wr.sum(0), 4), np.round(Hd.sum(1) / Hr.sum(1), 4),
                int((Wd != Wr).sum()), int((Hd != Hr).sum()), res[k]["iters"][r0], ref[k]["iters"][r0], res[k]["objvalue"][r0], ref[k]["objvalue"][r0]))
            print("rep", i, "k", k, "restarts", d, "maxdiff", max(float(np.abs(res[k]["W"][r] - ref[k]["W"][r]).max()) for r in d))
print("per k:", perk)
print("ITERS", iters, os.environ.get("NMFK_HIP_LIB", "new"), "reps", reps, "mismatching (rep,k) pairs:", bad)
