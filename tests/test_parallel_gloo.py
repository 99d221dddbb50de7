"""N > 1 path on CPU: two gloo ranks shard the restarts of every k (nmfk.jl_amd/parallel.py: plan_shards) and
exchange the results in one padded all-gather.  The compute function is injected (here: the CPU oracle, which only tests may use) so that the
partition / padding / gather logic is exercised without a GPU; on the GPU box the same code path runs with
Context.mu_sweep and the nccl (RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_sweep(X):
    import nmfk_oracle as oracle

    def sweep(ks, nruns, seeds=None, Winit=None, Hinit=None, params=None):
        n, m = X.shape
        out = {}
        for qi, k in enumerate(ks):
            W, H, obj, it, rs = [], [], [], [], []
            for r in range(nruns):
                W0, H0 = oracle.init_factors(int(seeds[qi][r]), n, m, k)
                res = oracle.singlerun(X, k, W0, H0, maxiter=int(params.maxiter), maxbaditers=int(params.maxbaditers))
                W.append(res["W"].astype(np.float32))
                H.append(res["H"].astype(np.float32))
                obj.append(np.float32(res["objvalue"]))
                it.append(res["iters"])
                rs.append(res["reason"])
            out[k] = dict(W=np.stack(W), H=np.stack(H), objvalue=np.array(obj, np.float32),
                          sse=np.array(obj, np.float64) ** 2, iters=np.array(it, np.int32), reason=np.array(rs, np.int32))
        return out

    return sweep


def _worker(rank, world, port, nruns, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import nmfk_jl_amd as NMFk
    import nmfk_oracle as oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n, m, ks = 12, 7, [2, 3]
        X0 = oracle.uniform_fill(3, 0, n * m).reshape(n, m).astype(np.float32) if rank == 0 else None
        X = NMFk.parallel.broadcast_X(X0)  # rank 0 owns X; everybody else receives it
        assert X.shape == (n, m)
        seeds = np.array([[NMFk.run_seed(5, k, r) for r in range(nruns)] for k in ks], dtype=np.uint64)
        params = NMFk.default_params(maxiter=30, maxbaditers=10 ** 9)
        res = NMFk.parallel.sharded_sweep(_oracle_sweep(X), ks, nruns, seeds, None, None, params, n, m)
        lean = NMFk.parallel.sharded_sweep(_oracle_sweep(X), ks, nruns, seeds, None, None, params, n, m, need_all_W=False)
        for k in ks:  # best-only exchange: H/objective complete, W of the best restart present on every rank
            best = int(np.argsort(res[k]["objvalue"], kind="stable")[0])
            assert np.array_equal(lean[k]["H"], res[k]["H"]) and np.array_equal(lean[k]["objvalue"], res[k]["objvalue"])
            assert np.array_equal(lean[k]["W"][best], res[k]["W"][best])
            _, chunks = NMFk.parallel.plan_shards(ks, nruns, world)
            own = {r for q, rs, g in chunks if g == rank and ks[q] == k for r in rs}
            assert len(own) in (nruns // world, -(-nruns // world))
            for r in range(nruns):
                if r in own:
                    assert np.array_equal(lean[k]["W"][r], res[k]["W"][r])
                elif r != best:
                    assert lean[k]["W"][r] is None
        q.put((rank, X, {k: {kk: np.asarray(v) for kk, v in res[k].items()} for k in ks}))
    finally:
        dist.destroy_process_group()


def test_plan_shards_covers_every_unit_once():
    import nmfk_jl_amd as NMFk

    for ks, nruns, world in [([2, 3, 4], 32, 8), ([5], 7, 4), ([2, 9], 2, 3), (list(range(2, 17)), 32, 1)]:
        c, chunks = NMFk.parallel.plan_shards(ks, nruns, world)
        assert c == -(-nruns // world)
        seen = sorted((q, r) for q, rs, g in chunks for r in rs)
        assert seen == [(q, r) for q in range(len(ks)) for r in range(nruns)]
        for q, rs, g in chunks:
            assert 0 <= g < world and 0 < len(rs) <= c
        per_rank = [sum(len(rs) for q, rs, g in chunks if g == r) for r in range(world)]
        assert max(per_rank) - min(per_rank) <= len(ks)


# nruns 5: uneven shards (rank 0 gets 3 restarts, rank 1 gets 2 + one padding run); world 3 with 2 restarts: an idle rank
@pytest.mark.parametrize("nruns,world", [(4, 2), (5, 2), (2, 3)])
def test_sharded_sweep_two_ranks(oracle, nruns, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nruns, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    (_, X0, r0), (_, X1, r1) = got[0], got[-1]
    np.testing.assert_array_equal(X0, X1)
    # every rank holds the complete, identically ordered result ...
    for k in r0:
        for key in r0[k]:
            np.testing.assert_array_equal(r0[k][key], r1[k][key])
            assert r0[k][key].shape[0] == nruns
    # ... equal to the unsharded computation
    import nmfk_jl_amd as NMFk

    ks = [2, 3]
    seeds = np.array([[NMFk.run_seed(5, k, r) for r in range(nruns)] for k in ks], dtype=np.uint64)
    params = NMFk.default_params(maxiter=30, maxbaditers=10 ** 9)
    ref = _oracle_sweep(X0)(ks, nruns, seeds=seeds, params=params)
    for k in ks:
        for key in ("W", "H", "objvalue", "iters", "reason"):
            np.testing.assert_array_equal(r0[k][key], ref[k][key])


def test_single_process_passthrough(oracle):
    import nmfk_jl_amd as NMFk

    assert NMFk.parallel.world() == (0, 1)
    X = oracle.uniform_fill(1, 0, 20).reshape(5, 4).astype(np.float32)
    np.testing.assert_array_equal(NMFk.parallel.broadcast_X(X), X)
    seeds = np.array([[1, 2]], dtype=np.uint64)
    params = NMFk.default_params(maxiter=10)
    res = NMFk.parallel.sharded_sweep(_oracle_sweep(X), [2], 2, seeds, None, None, params, 5, 4)
    assert res[2]["W"].shape == (2, 5, 2)


# ---------------------------------------------------------------------------------------------------------
# execute_run's solution filters on the GATHERED results of two ranks (Exec:552-596, 640-658), incl. the lean exchange
# (need_all_W=False: a rank holds the W of its own restarts and of the best one only).  The clustering / silhouette /
# fit calls of the post-processing go to a stand-in context built on the CPU oracle (tests may use it); on the GPU box
# the same host code runs with the real Context (tests/test_gpu_branches.py).
# ---------------------------------------------------------------------------------------------------------
class _OracleCtx:
    nan_count = 0

    def __init__(self, oracle, X):
        self.o, self.X = oracle, X

    def cluster_silhouette(self, Hs):
        labels, cent = self.o.clustersolutions(list(Hs), 32)
        _, ps, cs = self.o.finalize_silhouettes(list(Hs), labels, 32)
        return labels, cent, ps, cs

    def cluster_stats(self, Ws, Hs, labels):
        return self.o.cluster_stats(list(Ws), list(Hs), labels)

    def frobenius(self, W, H):
        return self.o.frobenius(self.X, W, H)


_FILTER_CASES = [dict(acceptratio=0.5), dict(acceptfactor=1.5), dict(best=False), dict(best=False, acceptratio=0.75),
                 dict(nanaction="removed")]


def _filter_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    import torch.distributed as dist
    import nmfk_jl_amd as NMFk
    import nmfk_oracle as oracle

    E = sys.modules[NMFk.execute.__module__]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n, m, k, R = 24, 10, 3, 6
        W0 = oracle.uniform_fill(71, 0, n * k).reshape(n, k)
        H0 = oracle.uniform_fill(72, 0, k * m).reshape(k, m)
        X = (W0 @ H0 + 0.05 * oracle.uniform_fill(73, 0, n * m).reshape(n, m)).astype(np.float32)
        seeds = np.array([[NMFk.run_seed(9, k, r) for r in range(R)]], dtype=np.uint64)
        params = NMFk.default_params(maxiter=80, maxbaditers=10 ** 9)
        out = []
        for opts in _FILTER_CASES:
            lean = opts.get("best", True)  # Exec:655-658: best=true needs the W of the best restart only
            res = NMFk.parallel.sharded_sweep(_oracle_sweep(X), [k], R, seeds, None, None, params, n, m, need_all_W=not lean)[k]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                Wa, Ha, phi, sil, aic, extra = E._execute_run_post(_OracleCtx(oracle, X), X, k, R, res, **opts)
            out.append(dict(Wa=np.asarray(Wa), Ha=np.asarray(Ha), phi=phi, sil=sil, aic=aic, labels=np.asarray(extra["labels"]),
                            idxsort=np.asarray(extra["idxsort"])))
        q.put((rank, X, out))
    finally:
        dist.destroy_process_group()


def test_execute_run_filters_on_two_ranks(oracle):
    import nmfk_jl_amd as NMFk

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_filter_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, X, a), (_, _, b) = got
    n, m = X.shape
    k, R = 3, 6
    inits = [oracle.init_factors(NMFk.run_seed(9, k, r), n, m, k) for r in range(R)]
    for opts, ra, rb in zip(_FILTER_CASES, a, b):
        for key in ra:  # every rank reaches the same answer ...
            np.testing.assert_array_equal(np.asarray(ra[key]), np.asarray(rb[key]))
        ref = oracle.execute_run(X, k, R, inits, maxiter=80, maxbaditers=10 ** 9, **opts)  # ... the unsharded one
        assert (ra["idxsort"] == ref["idxsort"]).all() and (ra["labels"] == ref["labels"]).all(), opts
        assert abs(ra["phi"] - ref["phi"]) <= 1e-5 * ref["phi"] and abs(ra["sil"] - ref["minsilhouette"]) <= 1e-5
        np.testing.assert_allclose(ra["Wa"] @ ra["Ha"], ref["Wa"] @ ref["Ha"], rtol=1e-4, atol=1e-6)
