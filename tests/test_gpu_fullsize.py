"""GPU parity at the METRIC's own size (BASELINE.json: 8192 x 512, k = 2:16, nruns = 32), through the C ABI.

  * the bench's dominant schedule (two phases: ranks 9..16 as one split-operand MFMA group of 256 factorizations, then
    the small ranks on the packed-VALU kernels) against the Float64 oracle on sampled units, fixed budget;
  * "same kopt": SURVEY 8d's planted rank-6 matrix, default stop rule -> kopt = 6 in fp32 compute, the same kopt and
    the same set of ranks above the cutoff as the fp64 compute mode (whose stop decisions are oracle-verified by
    test_stop_rule_fp64_identical_iterations), reference rule src/NMFkExecute.jl:225, src/NMFkPostprocess.jl:7-41."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NOSTOP = dict(maxbaditers=10 ** 9)


@pytest.fixture(scope="module")
def NMFk():
    import nmfk_jl_amd

    return nmfk_jl_amd


@pytest.fixture(scope="module")
def ctx(NMFk):
    c = NMFk.Context(0)
    yield c
    c.close()


def _rel(a, b, X):
    return np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(X)


def planted_X(ctx, n=8192, m=512, k0=6, seed=2):
    """X = W0*H0 + 0.01*U, W0 in U(0,1)^{n x 6}, H0 in U(0,1)^{6 x m}, from the library's
    portable generator so that the matrix is identical everywhere."""
    W0 = ctx.fill_uniform(seed, 0, n * k0).reshape(k0, n).T.astype(np.float64)
    H0 = ctx.fill_uniform(seed, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
    U = ctx.fill_uniform(seed, n * k0 + k0 * m, n * m).reshape(m, n).T.astype(np.float64)
    return np.asfortranarray((W0 @ H0 + 0.01 * U).astype(np.float32))


def test_bench_schedule_fixed_budget_vs_oracle(NMFk, ctx, oracle):
    """The whole bench sweep (15 ranks x 32 restarts) for 20 iterations under the DEFAULT schedule; asserts that the
    two-phase schedule with the 256-unit MFMA group was taken, then compares sampled units of both phases with
    oracle.singlerun from identical initial factors (fp32 tolerance 1e-4)."""
    n, m = 8192, 512
    X = np.asfortranarray(ctx.fill_uniform(1, 0, n * m).reshape(m, n).T)
    ctx.set_X(X)
    ks, R, iters = list(range(2, 17)), 32, 20
    seeds = np.array([[NMFk.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    res = ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, **NOSTOP)
    info = ctx.last_sweep_info()
    assert info["phases"] == 2 and info["mfma_group_units"] == 8 * R, info
    for k, r in [(16, 31), (9, 0), (12, 17), (8, 31), (2, 5), (5, 16)]:
        W0, H0 = oracle.init_factors(int(seeds[k - 2, r]), n, m, k)
        ref = oracle.singlerun(X, k, W0, H0, maxiter=iters, nthreads=8, **NOSTOP)
        assert res[k]["iters"][r] == iters
        assert _rel(res[k]["W"][r] @ res[k]["H"][r], ref["W"] @ ref["H"], X) <= 1e-4, (k, r)
        assert abs(res[k]["objvalue"][r] - ref["objvalue"]) <= 1e-4 * ref["objvalue"], (k, r)
        np.testing.assert_allclose(res[k]["H"][r].sum(axis=1), 1.0, atol=1e-4)


def test_planted_rank6_same_kopt_at_metric_size(NMFk, ctx):
    """execute(X, 2:16, 32) on the planted rank-6 8192x512 matrix, default schedule and stop rule: kopt = 6, and the
    fp64 compute mode (the reference's arithmetic and stop decisions) agrees on kopt, on which ranks pass the
    cutoff, and on the fit of every rank to 1 %."""
    X = planted_X(ctx)
    ctx.set_X(X)
    out = {}
    for mode in ("f32", "f64"):
        W, H, fit, rob, aic, kopt = NMFk.execute(X, range(2, 17), 32, load=False, save=False, quiet=True, seed=2, ctx=ctx,
                                                 compute=mode)
        out[mode] = (np.array(fit), np.array(rob), kopt)
        if mode == "f32":
            assert ctx.last_sweep_info()["phases"] == 2
    (fit32, rob32, k32), (fit64, rob64, k64) = out["f32"], out["f64"]
    assert k32 == 6 and k64 == 6
    assert ((rob32[1:] > 0.5) == (rob64[1:] > 0.5)).all()
    np.testing.assert_allclose(fit32[1:], fit64[1:], rtol=1e-2)
    assert rob32[5] > 0.9 and rob64[5] > 0.9
