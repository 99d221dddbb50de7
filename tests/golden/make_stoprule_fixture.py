#!/usr/bin/env python3
"""Generates tests/golden/stoprule_planted_1024x256.npz from the CPU oracle (run ONCE in the build container:
`python tests/golden/make_stoprule_fixture.py [workers]`; about 20 minutes on 6 cores).

What it pins (round-2 verdict, "close the parity chain at the point the bench lives on"): the reference's DEFAULT stop
rule (src/NMFkMultiplicative.jl:73-98: tolOF = 1e-3 on the objective monitored every 10th iteration, maxbaditers = 10,
maxreattempts = 2, maxiter = 10000, src/NMFkExecute.jl:729) driving a whole `execute` (src/NMFkExecute.jl:178-233) on a
matrix large enough that the GPU library takes its matrix-pipe (split-operand MFMA) schedule with the MFMA objective --
the kernels the bench spends its time in.  X = W0*H0 + 0.01*U with a planted rank of 5, 1024 x 256, Float32; k = 2:13
(every kernel variant: 1-3 bf16 MFMAs for W*H, 1-4 blocks of four signals for the numerators), 16 restarts per k.

Stored (data only -- inputs are regenerated from the portable generator by seed):
  fit, robustness, aic per k, kopt                                  the outputs of execute
  objvalue, iters, reason per (k, restart)                          Exec:529-531, Mult:64/75-78/112-115
  trace_k<k>_r<r>                                                   the monitored objective (Mult:74) at every check of
                                                                    three restarts (k = 3, 8, 13; restart 0)
The fixture is compared with libnmfk_hip by tests/test_gpu_fullsize.py::test_default_stop_rule_against_the_oracle_fixture."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nmfk_oracle as oracle  # noqa: E402

N, M, K0, NOISE, XSEED = 1024, 256, 5, 0.01, 2
KS, NRUNS, SEED = list(range(2, 14)), 16, 2021
TRACES = [(3, 0), (8, 0), (13, 0)]


def planted_X():
    """the bench's planted construction (bench.py) at this size: columns of the generator's stream"""
    W0 = oracle.uniform_fill(XSEED, 0, N * K0).reshape(K0, N).T
    H0 = oracle.uniform_fill(XSEED, N * K0, K0 * M).reshape(M, K0).T
    U = oracle.uniform_fill(XSEED, N * K0 + K0 * M, N * M).reshape(M, N).T
    return np.asfortranarray((W0 @ H0 + NOISE * U).astype(np.float32))


def one_rank(nk):
    X = planted_X()
    inits = [oracle.init_factors(oracle.run_seed(SEED, nk, r), N, M, nk) for r in range(NRUNS)]
    t = time.perf_counter()
    r = oracle.execute_run(X, nk, NRUNS, inits)
    so = oracle.signalorder(r["Wa"], r["Ha"])
    W, H = r["Wa"][:, so], r["Ha"][so, :]
    fit = oracle.normnan(np.asarray(X, dtype=np.float64) - W.astype(np.float32) @ H.astype(np.float32))  # Exec:211-222
    traces = {}
    for (k, rr) in TRACES:
        if k == nk:
            tr = oracle.multiplicative(X, nk, inits[rr][0], inits[rr][1], trace=True)
            assert tr["iters"] == r["iters"][rr]
            traces[f"trace_k{k}_r{rr}"] = tr["trace"]
    return dict(nk=nk, fit=fit, rob=r["minsilhouette"], aic=r["aic"], objvalue=np.asarray(r["objvalue"], dtype=np.float32),
                iters=np.asarray(r["iters"], dtype=np.int64), reason=np.asarray(r["reasons"], dtype=np.int32), traces=traces,
                seconds=time.perf_counter() - t)


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    oracle.build()
    t0 = time.perf_counter()
    with ProcessPoolExecutor(workers) as ex:
        res = list(ex.map(one_rank, sorted(KS, reverse=True)))  # long ranks first
    res.sort(key=lambda d: d["nk"])
    rob = np.array([d["rob"] for d in res], dtype=np.float32)
    kopt = oracle.getk(KS, rob, 0.5)
    out = dict(n=N, m=M, k0=K0, noise=NOISE, xseed=XSEED, ks=np.array(KS), nruns=NRUNS, seed=SEED,
               fit=np.array([d["fit"] for d in res], dtype=np.float32), robustness=rob,
               aic=np.array([d["aic"] for d in res], dtype=np.float32), kopt=kopt,
               objvalue=np.stack([d["objvalue"] for d in res]), iters=np.stack([d["iters"] for d in res]),
               reason=np.stack([d["reason"] for d in res]))
    for d in res:
        out.update(d["traces"])
    path = os.path.join(ROOT, "tests", "golden", f"stoprule_planted_{N}x{M}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} in {time.perf_counter() - t0:.0f} s: kopt = {kopt}")
    for d in res:
        print(f"  k={d['nk']:2d} fit {d['fit']:.5f} robustness {d['rob']: .4f} iters {d['iters'].min()}..{d['iters'].max()} "
              f"reasons {sorted(set(d['reason'].tolist()))} ({d['seconds']:.0f} s)")


if __name__ == "__main__":
    main()
