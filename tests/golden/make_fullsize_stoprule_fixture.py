#!/usr/bin/env python3
"""Generates tests/golden/stoprule_fullsize_8192x512.npz from the CPU oracle (run ONCE in the build container:
`python tests/golden/make_fullsize_stoprule_fixture.py [workers]`; five independent restarts, ~20 minutes on 5 cores).

What it pins (round-5 verdict, "close parity at the metric's own size with an oracle fixture"): the reference's DEFAULT
stop rule (src/NMFkMultiplicative.jl:64-117: tolOF = 1e-3 on the objective monitored every 10th iteration, maxbaditers =
10, maxreattempts = 2, maxiter = 10000 from src/NMFkExecute.jl:729) at BASELINE.json's own shape, 8192 x 512 Float32:

  * SURVEY 8d cfg3 (ii), the planted rank-6 matrix X = W0*H0 + 0.01*U (seed 2): restart 0 of k = 3, 6, 12 (one rank per
    first-product form of the matrix-pipe kernels), seeds as `execute(X, 2:16, 32; seed = 2)` derives them;
  * SURVEY 8d cfg3 (i), the headline noise matrix U(0,1) (seed 1): restart 0 of k = 2 and k = 16, seeds as bench.py
    derives them (they run to maxiter: the trace is the point).

Stored per restart (data only -- inputs are regenerated from the portable generator by seed):
  trace_<tag>_k<k>     the monitored objective (Mult:74) at every check
  iters, reason, sse, objvalue (Frobenius objective of execute_singlerun, Exec:790-805)
The fixture is compared with libnmfk_hip by
tests/test_gpu_fullsize.py::test_default_stop_rule_at_metric_size_against_the_oracle_fixture, where the five units are
followed by uid INSIDE the bench's own 480-unit default-schedule sweep."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nmfk_oracle as oracle  # noqa: E402

N, M, K0, NOISE = 8192, 512, 6, 0.01
CASES = [("noise", 1, 16, 0), ("planted", 2, 12, 0), ("planted", 2, 6, 0), ("planted", 2, 3, 0), ("noise", 1, 2, 0)]  # (tag, seed, k, restart)


def matrix(tag, seed):
    if tag == "noise":  # bench.py / tests/test_gpu_fullsize.py: fill_uniform(1, 0, n*m).reshape(m, n).T
        return np.asfortranarray(oracle.uniform_fill(seed, 0, N * M).reshape(M, N).T.astype(np.float32))
    W0 = oracle.uniform_fill(seed, 0, N * K0).reshape(K0, N).T
    H0 = oracle.uniform_fill(seed, N * K0, K0 * M).reshape(M, K0).T
    U = oracle.uniform_fill(seed, N * K0 + K0 * M, N * M).reshape(M, N).T
    return np.asfortranarray((W0 @ H0 + NOISE * U).astype(np.float32))


def one(case):
    tag, seed, k, r = case
    nthreads = int(os.environ.get("FIXTURE_THREADS", "1"))
    X = matrix(tag, seed)
    W0, H0 = oracle.init_factors(oracle.run_seed(seed, k, r), N, M, k)
    t = time.perf_counter()
    tr = oracle.multiplicative(X, k, W0, H0, trace=True, nthreads=nthreads)
    objvalue = oracle.frobenius(X, tr["W"], tr["H"])  # execute_singlerun_compute's objective (Exec:790-805) of the same factors
    return dict(tag=tag, seed=seed, k=k, r=r, trace=tr["trace"], iters=tr["iters"], reason=tr["reason"], sse=tr["sse"],
                objvalue=objvalue, seconds=time.perf_counter() - t)


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    oracle.build()
    t0 = time.perf_counter()
    with ProcessPoolExecutor(workers) as ex:
        res = list(ex.map(one, CASES))
    out = dict(n=N, m=M, k0=K0, noise=NOISE, tags=np.array([c[0] for c in CASES]), seeds=np.array([c[1] for c in CASES]),
               ks=np.array([c[2] for c in CASES]), restarts=np.array([c[3] for c in CASES]),
               iters=np.array([d["iters"] for d in res], dtype=np.int64), reason=np.array([d["reason"] for d in res], dtype=np.int32),
               sse=np.array([d["sse"] for d in res]), objvalue=np.array([d["objvalue"] for d in res]))
    for d in res:
        out[f"trace_{d['tag']}_k{d['k']}"] = d["trace"]
    path = os.path.join(ROOT, "tests", "golden", f"stoprule_fullsize_{N}x{M}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} in {time.perf_counter() - t0:.0f} s")
    for d in res:
        print(f"  {d['tag']:8s} k={d['k']:2d} r={d['r']} iters {d['iters']} reason {d['reason']} checks {len(d['trace'])} "
              f"objective {d['trace'][0]:.6g} -> {d['trace'][-1]:.6g} objvalue {d['objvalue']:.6g} ({d['seconds']:.0f} s)")


if __name__ == "__main__":
    main()
