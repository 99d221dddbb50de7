"""The multi-GPU entry points of the C ABI (nmfk_comm_*, nmfk_mu_sweep_sharded, nmfk_multi_*) on the one GPU the test box
has: a communicator of ONE rank runs the real RCCL calls (ncclCommInitRank, ncclBroadcast of X, ncclAllGather of the
device result buffers, the strided delivery) and must reproduce the plain sweep bit for bit.  The sharding arithmetic
for N > 1 is covered on the CPU (tests/test_parallel_gloo.py, tests/test_host_cpu.py::test_shard_plan_*)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NOSTOP = dict(maxbaditers=10 ** 9)


@pytest.fixture(scope="module")
def NMFk():
    import nmfk_jl_amd

    return nmfk_jl_amd


def _case(oracle, n=300, m=70):
    return (0.05 + oracle.uniform_fill(91, 0, n * m)).reshape(n, m).astype(np.float32)


def test_comm_world_size_one_matches_plain_sweep(NMFk, oracle):
    from nmfk_jl_amd import _lib

    X = _case(oracle)
    ks, R = [2, 5, 12, 20], 5
    seeds = np.array([[NMFk.run_seed(3, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref_ctx = NMFk.Context(0)
    ref_ctx.set_X(X)
    ref = ref_ctx.mu_sweep(ks, R, seeds=seeds, maxiter=30, **NOSTOP)
    ctx = NMFk.Context(0)
    comm = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
    comm.bcast_X(X, root=0)  # ncclBroadcast + NMFpreprocessing! on the device copy
    assert (ctx.n, ctx.m) == X.shape and ctx.nan_count == 0
    for need_W in (True, False):
        res = comm.mu_sweep(ks, R, seeds=seeds, maxiter=30, need_W=need_W, **NOSTOP)
        for k in ks:
            for key in ("W", "H", "objvalue", "iters", "reason", "sse"):
                assert (res[k][key] == ref[k][key]).all(), (k, key, need_W)
    # given initial factors travel through the shard buffers too
    k = 3
    W0 = np.stack([oracle.init_factors(int(s), *X.shape, k)[0] for s in seeds[0]])
    H0 = np.stack([oracle.init_factors(int(s), *X.shape, k)[1] for s in seeds[0]])
    a = comm.mu_sweep([k], R, Winit={k: W0}, Hinit={k: H0}, maxiter=20, **NOSTOP)[k]
    b = ref_ctx.mu_sweep([k], R, Winit={k: W0}, Hinit={k: H0}, maxiter=20, **NOSTOP)[k]
    assert (a["W"] == b["W"]).all() and (a["H"] == b["H"]).all()
    comm.close()
    ctx.close()
    ref_ctx.close()


def test_multi_single_gpu_and_execute_through_comm(NMFk, oracle):
    """nmfk_multi_* with one GPU (threads + communicator + delivery through GPU 0), and execute() on a context that is
    attached to a communicator: same results as the unattached path."""
    from nmfk_jl_amd import _lib, parallel

    X = _case(oracle, 96, 24)
    ks, R = [2, 3, 4], 6
    seeds = np.array([[NMFk.run_seed(7, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref_ctx = NMFk.Context(0)
    ref_ctx.set_X(X)
    ref = ref_ctx.mu_sweep(ks, R, seeds=seeds, maxiter=40, **NOSTOP)
    mh = _lib.Multi(1)
    mh.set_X(X)
    res = mh.mu_sweep(ks, R, seeds=seeds, maxiter=40, **NOSTOP)
    for k in ks:
        assert (res[k]["W"] == ref[k]["W"]).all() and (res[k]["H"] == ref[k]["H"]).all()
        assert (res[k]["objvalue"] == ref[k]["objvalue"]).all()
    assert abs(mh.ctx0.frobenius(res[2]["W"][0], res[2]["H"][0]) - ref_ctx.frobenius(ref[2]["W"][0], ref[2]["H"][0])) < 1e-6
    mh.close()
    out_ref = NMFk.execute(X, ks, R, load=False, save=False, quiet=True, seed=5, ctx=ref_ctx, maxiter=200)
    ctx = NMFk.Context(0)
    comm = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
    comm.bcast_X(X)
    parallel._comms[id(ctx)] = comm
    try:
        out = NMFk.execute(X, ks, R, load=False, save=False, quiet=True, seed=5, ctx=ctx, maxiter=200)
    finally:
        parallel.detach(ctx)
    assert out[5] == out_ref[5]
    np.testing.assert_array_equal(out[3], out_ref[3])
    for k in ks:
        np.testing.assert_array_equal(out[0][k - 1], out_ref[0][k - 1])
    ctx.close()
    ref_ctx.close()


def test_comm_errors_are_loud(NMFk):
    from nmfk_jl_amd import _lib

    ctx = NMFk.Context(0)
    with pytest.raises(NMFk.NMFkError):
        _lib.Comm(ctx, 2, 5, bytes(128))  # rank out of range
    comm = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
    with pytest.raises(NMFk.NMFkError, match="nmfk_set_X"):
        comm.mu_sweep([2], 2, seeds=np.zeros((1, 2), np.uint64))
    comm.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 through the C ABI on ONE GPU: the loopback transport (include/nmfk_hip.h, "Loopback transport").  N logical ranks
# = N contexts + N host threads; everything above the three collective primitives -- shard plan, padding by repeating the
# last restart, contribution layout, strided delivery (pitch elem * N), the need_W = 0 local-W copy, the status agreement
# and the thread fan-out of nmfk_multi_* -- is the code the RCCL transport runs.  Reference seam: Distributed.pmap over the
# restarts, src/NMFkExecute.jl:511-526.
# ---------------------------------------------------------------------------------------------------------------------
def _ranks_in_threads(mh, fn):
    """fn(g, comm) on one Python thread per logical rank (ctypes releases the GIL, the collectives meet inside the library)."""
    import threading

    out, err = [None] * mh.ngpus, [None] * mh.ngpus

    def work(g):
        try:
            out[g] = fn(g, mh.comm(g))
        except Exception as e:  # noqa: BLE001 -- handed to the asserting thread
            err[g] = e

    th = [threading.Thread(target=work, args=(g,)) for g in range(mh.ngpus)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a rank is still blocked in a collective"
    return out, err


def _shard_reference(NMFk, ref_ctx, ks, R, N, seeds=None, Winit=None, Hinit=None, **kw):
    """What the sharded sweep must deliver, bit for bit: rank g runs nmfk_mu_sweep on ITS restarts {g, g + N, ...} padded to
    ceil(R / N) by repeating the last one (the launch schedule of a sweep depends on the number of restarts per rank, so
    the comparison is against the same shard swept alone, not against the unsharded sweep -- that one is compared within
    fp32 tolerance).  -> dict k -> dict key -> array over all R restarts"""
    from nmfk_jl_amd import _lib

    out = {k: {} for k in ks}
    for g in range(N):
        cnt, pad = _lib.shard_plan(R, N, g)
        if cnt == 0:
            continue
        idx = [g + min(j, cnt - 1) * N for j in range(pad)]
        part = ref_ctx.mu_sweep(ks, pad, seeds=None if seeds is None else seeds[:, idx],
                                Winit=None if Winit is None else {k: v[idx] for k, v in Winit.items()},
                                Hinit=None if Hinit is None else {k: v[idx] for k, v in Hinit.items()}, **kw)
        for k in ks:
            for key, val in part[k].items():
                dst = out[k].setdefault(key, np.zeros((R,) + val.shape[1:], dtype=val.dtype))
                dst[idx[:cnt]] = val[:cnt]
    return out


@pytest.mark.parametrize("N,R", [(2, 5), (3, 7), (8, 5), (8, 11), (3, 2)])
def test_loopback_sharded_sweep_is_bit_identical(NMFk, oracle, N, R):
    """nruns not divisible by N, more ranks than restarts (idle ranks), mixed rank widths incl. the all-MFMA kernel."""
    from nmfk_jl_amd import _lib

    X = _case(oracle, 200, 48)
    ks = [2, 5, 12, 20]
    seeds = np.array([[NMFk.run_seed(11, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref_ctx = NMFk.Context(0)
    ref_ctx.set_X(X)
    ref = _shard_reference(NMFk, ref_ctx, ks, R, N, seeds=seeds, maxiter=30, **NOSTOP)
    whole = ref_ctx.mu_sweep(ks, R, seeds=seeds, maxiter=30, **NOSTOP)
    for k in ks:  # the shards' results are the unsharded sweep's up to the schedule's summation order
        for r in range(R):
            err = np.linalg.norm(ref[k]["W"][r] @ ref[k]["H"][r] - whole[k]["W"][r] @ whole[k]["H"][r]) / np.linalg.norm(X)
            assert err < 1e-5, (k, r, err)
    mh = _lib.Multi(N, loopback=True)
    mh.set_X(X)  # loopback broadcast of X from rank 0 + NMFpreprocessing! on every rank
    res = mh.mu_sweep(ks, R, seeds=seeds, maxiter=30, **NOSTOP)  # nmfk_multi_sweep: N threads inside the library
    for k in ks:
        for key in ("W", "H", "objvalue", "iters", "reason", "sse"):
            assert (res[k][key] == ref[k][key]).all(), (k, key)

    # per-rank calls of nmfk_mu_sweep_sharded: every rank receives every restart; need_W = 0: W of the own restarts only
    for need_W in (True, False):
        out, err = _ranks_in_threads(mh, lambda g, comm: comm.mu_sweep(ks, R, seeds=seeds, maxiter=30, need_W=need_W, **NOSTOP))
        assert err == [None] * N, err
        for g in range(N):
            own = [r for r in range(R) if _lib.shard_owner(R, N, r)[0] == g]
            assert own == list(range(g, R, N))
            for k in ks:
                for key in ("H", "objvalue", "iters", "reason", "sse"):
                    assert (out[g][k][key] == ref[k][key]).all(), (g, k, key, need_W)
                if need_W:
                    assert (out[g][k]["W"] == ref[k]["W"]).all()
                else:
                    assert (out[g][k]["W"][own] == ref[k]["W"][own]).all()
                    other = [r for r in range(R) if r not in own]
                    assert np.isnan(out[g][k]["W"][other]).all()  # (the binding pre-fills W with NaN)
    mh.close()
    ref_ctx.close()


def test_loopback_sharded_sweep_on_sparse_X(NMFk, oracle):
    """BASELINE configs[3] data through the N > 1 code: sparse X is set on every rank's context (nmfk_set_X_csc takes host
    pointers; there is no broadcast for it), the sharded sweep delivers every restart, and the result of each restart is
    bit for bit what its shard gives swept alone on one context."""
    import scipy.sparse as sp
    from nmfk_jl_amd import _lib

    n, m, N, R = 400, 130, 3, 5
    pos = oracle.uniform_fill(61, 0, n * m).reshape(n, m) < 0.06
    X = np.where(pos, 1 + 4 * oracle.uniform_fill(62, 0, n * m).reshape(n, m), 0.0).astype(np.float32)
    Xs = sp.csc_matrix(X)
    ks = [3, 10, 20, 36]  # blocked form where the library takes it, gather form for k = 36
    seeds = np.array([[NMFk.run_seed(13, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ref_ctx = NMFk.Context(0)
    ref_ctx.set_X_sparse(Xs)
    ref = _shard_reference(NMFk, ref_ctx, ks, R, N, seeds=seeds, maxiter=25, **NOSTOP)
    mh = _lib.Multi(N, loopback=True)
    with pytest.raises(NMFk.NMFkError) as e:  # (no X yet: the binding refuses instead of handing out empty result buffers)
        mh.mu_sweep(ks, R, seeds=seeds, maxiter=25, **NOSTOP)
    assert e.value.code == 4
    mh.set_X_sparse(Xs)
    res = mh.mu_sweep(ks, R, seeds=seeds, maxiter=25, **NOSTOP)
    for k in ks:
        for key in ("W", "H", "objvalue", "iters", "reason"):
            assert (res[k][key] == ref[k][key]).all(), (k, key)
    mh.close()
    ref_ctx.close()


def test_loopback_given_inits_travel_through_the_shard_buffers(NMFk, oracle):
    from nmfk_jl_amd import _lib

    X = _case(oracle, 120, 40)
    N, R, k = 3, 7, 4
    seeds = [NMFk.run_seed(2, k, r) for r in range(R)]
    W0 = np.stack([oracle.init_factors(int(s), *X.shape, k)[0] for s in seeds])
    H0 = np.stack([oracle.init_factors(int(s), *X.shape, k)[1] for s in seeds])
    ref_ctx = NMFk.Context(0)
    ref_ctx.set_X(X)
    b = _shard_reference(NMFk, ref_ctx, [k], R, N, Winit={k: W0}, Hinit={k: H0}, maxiter=25, **NOSTOP)[k]
    mh = _lib.Multi(N, loopback=True)
    mh.set_X(X)
    a = mh.mu_sweep([k], R, Winit={k: W0}, Hinit={k: H0}, maxiter=25, **NOSTOP)[k]
    assert (a["W"] == b["W"]).all() and (a["H"] == b["H"]).all() and (a["objvalue"] == b["objvalue"]).all()
    # only H given: W drawn from the seeds (the mixed form of Mult:38-55)
    sd = np.array([seeds], dtype=np.uint64)
    a = mh.mu_sweep([k], R, seeds=sd, Hinit={k: H0}, maxiter=25, **NOSTOP)[k]
    b = _shard_reference(NMFk, ref_ctx, [k], R, N, seeds=sd, Hinit={k: H0}, maxiter=25, **NOSTOP)[k]
    assert (a["W"] == b["W"]).all() and (a["H"] == b["H"]).all()
    mh.close()
    ref_ctx.close()


def test_loopback_execute_with_the_lean_W_exchange(NMFk, oracle):
    """execute() on a communicator with best = true (the default): the sharded sweep runs with need_W = 0 -- H, objective,
    iterations of every restart travel, W only stays with its owner -- and the owner of the restart with the lowest objective
    (Exec:545-546) hands its W to the others (nmfk_comm_bcast; SURVEY 8e "send of the winning W").  Three logical ranks on
    the loopback transport, one Python thread each: every rank returns the same sweep result, equal to the one-context
    execute() up to the shards' summation order; best = false takes the full exchange and agrees as well."""
    from nmfk_jl_amd import _lib, parallel

    n, m, k0, N, R = 150, 40, 3, 3, 5
    W0 = oracle.uniform_fill(81, 0, n * k0).reshape(n, k0)
    H0 = oracle.uniform_fill(82, 0, k0 * m).reshape(k0, m)
    X = np.asfortranarray((W0 @ H0 + 0.01 * oracle.uniform_fill(83, 0, n * m).reshape(n, m)).astype(np.float32))
    ks = [2, 3, 4]
    ref_ctx = NMFk.Context(0)
    ref_ctx.set_X(X)
    kw = dict(load=False, save=False, quiet=True, seed=7, maxiter=300)
    refs = {best: NMFk.execute(X, ks, R, ctx=ref_ctx, best=best, **kw) for best in (True, False)}
    mh = _lib.Multi(N, loopback=True)
    mh.set_X(X)
    for best in (True, False):
        ref = refs[best]
        bcasts = [0] * N

        def work(g, comm):
            parallel._comms[id(comm.ctx)] = comm
            inner = comm.bcast

            def counting(arr, root):
                bcasts[g] += 1
                return inner(arr, root)

            comm.bcast = counting
            try:
                return NMFk.execute(X, ks, R, ctx=comm.ctx, best=best, **kw)
            finally:
                parallel._comms.pop(id(comm.ctx), None)

        out, err = _ranks_in_threads(mh, work)
        assert err == [None] * N, err
        assert bcasts == ([len(ks)] * N if best else [0] * N), bcasts  # one winning W per rank k, or the full exchange
        for g in range(N):
            W, H, fit, rob, aic, kopt = out[g]
            assert kopt == ref[5] == 3
            for k in ks:
                if best:
                    assert (W[k - 1] == out[0][0][k - 1]).all() and (H[k - 1] == out[0][1][k - 1]).all()  # every rank the same bits
                    assert np.linalg.norm(W[k - 1] @ H[k - 1] - ref[0][k - 1] @ ref[1][k - 1]) <= 1e-4 * np.linalg.norm(X)
                np.testing.assert_allclose(fit[k - 1], ref[2][k - 1], rtol=1e-3)  # (best = false: the fit of the cluster MEANS)
                np.testing.assert_allclose(rob[k - 1], out[0][3][k - 1], atol=1e-6)
            np.testing.assert_allclose(np.array(rob)[[k - 1 for k in ks]], np.array(ref[3])[[k - 1 for k in ks]], atol=2e-3)
    mh.close()
    ref_ctx.close()


def test_loopback_failing_rank_fails_every_rank_without_a_hang(NMFk, oracle):
    """A NaN initial factor in ONE shard (restart 1 -> rank 1 of 3): that rank's local sweep returns NMFK_ERR_NAN_INIT; the
    status agreement makes every rank (and nmfk_multi_sweep) return it instead of blocking in the all-gather."""
    from nmfk_jl_amd import _lib

    X = _case(oracle, 120, 40)
    N, R, k = 3, 6, 3
    seeds = [NMFk.run_seed(4, k, r) for r in range(R)]
    W0 = np.stack([oracle.init_factors(int(s), *X.shape, k)[0] for s in seeds])
    H0 = np.stack([oracle.init_factors(int(s), *X.shape, k)[1] for s in seeds])
    W0[1, 5, 1] = np.nan
    mh = _lib.Multi(N, loopback=True)
    mh.set_X(X)
    with pytest.raises(NMFk.NMFkError, match="GPU 1: Initial values") as ei:
        mh.mu_sweep([k], R, Winit={k: W0}, Hinit={k: H0}, maxiter=10, **NOSTOP)
    assert ei.value.code == _lib.ERR_NAN_INIT
    out, err = _ranks_in_threads(mh, lambda g, comm: comm.mu_sweep([k], R, Winit={k: W0}, Hinit={k: H0}, maxiter=10, **NOSTOP))
    assert all(isinstance(e, NMFk.NMFkError) and e.code == _lib.ERR_NAN_INIT for e in err), err
    assert "Initial values" in str(err[1]) and "rank 1 of 3 failed" in str(err[0]) and "rank 1 of 3 failed" in str(err[2])
    # the communicators are still usable afterwards
    W0[1, 5, 1] = 0.5
    a = mh.mu_sweep([k], R, Winit={k: W0}, Hinit={k: H0}, maxiter=10, **NOSTOP)[k]
    assert np.isfinite(a["objvalue"]).all()
    # X with a negative entry: the root's NMFpreprocessing! fails on every rank alike -> one error, no hang
    Xbad = X.copy()
    Xbad[3, 3] = -1.0
    with pytest.raises(NMFk.NMFkError, match="nonnegative"):
        mh.set_X(Xbad)
    mh.close()
