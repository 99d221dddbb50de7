"""Do the packed-VALU kernels give bit-identical results while MFMA kernels of ANOTHER context run on the same GPU?"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nmfk_jl_amd as NMFk, nmfk_oracle as oracle
n, m = 700, 130
X = (0.05 + oracle.uniform_fill(33, 0, n * m)).reshape(n, m).astype(np.float32)
NOSTOP = dict(maxbaditers=10 ** 9)
mode = sys.argv[1] if len(sys.argv) > 1 else "hyb"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
A = NMFk.Context(0); A.set_X(X)
B = NMFk.Context(0); B.set_X(X)
ksA, R = [int(v) for v in os.environ.get("KSA", "2,3,5").split(",")], int(os.environ.get("RA", "8"))
seedsA = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ksA], dtype=np.uint64)
stop = False
def burn():
    ksB = {"hyb": [13, 16, 9, 12], "wide": [20, 32], "valu": [2, 3, 4, 5]}[mode]
    RB = 4 if mode != "valu" else 4
    seedsB = np.array([[NMFk.run_seed(5, k, r) for r in range(RB)] for k in ksB], dtype=np.uint64)
    env_iters = int(os.environ.get("BITERS", "40"))
    while not stop:
        B.mu_sweep(ksB, RB, seeds=seedsB, maxiter=env_iters, **NOSTOP)
extra = dict(compute=NMFk.COMPUTE_F64) if os.environ.get("AF64") else {}
ref = A.mu_sweep(ksA, R, seeds=seedsA, maxiter=40, **NOSTOP, **extra)
for phase in ("alone", "with " + mode):
    th = None
    if phase != "alone":
        th = threading.Thread(target=burn); th.start(); time.sleep(0.2)
    bad = 0
    for i in range(reps):
        res = A.mu_sweep(ksA, R, seeds=seedsA, maxiter=40, **NOSTOP, **extra)
        for k in ksA:
            if not ((res[k]["W"] == ref[k]["W"]).all() and (res[k]["H"] == ref[k]["H"]).all()):
                bad += 1
    if th:
        stop = True; th.join()
    print(phase, "reps", reps, "mismatching (rep,k):", bad, "info", A.last_sweep_info())
