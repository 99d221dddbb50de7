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
