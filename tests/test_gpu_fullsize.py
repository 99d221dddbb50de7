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


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at its own size: sparse 0.5 %-fill 100000 x 4096 through the CSC/CSR gather kernels
# ---------------------------------------------------------------------------------------------------------
def _gather_form_mu(Xs, W, H, iters):
    """The reference half-steps (Mult:67,70) restricted to the stored non-zeros, Float64, scipy.sparse: a zero of X is
    lambda = 1e-32 in the reference (Mult:17-18), its ratio X/(W*H) ~ 1e-32 contributes nothing.  Returns W, H and the
    monitored objective norm(X - W*H) over ALL entries (Mult:74)."""
    import scipy.sparse as sp

    Xs = sp.csc_matrix(Xs).astype(np.float64)
    Xs.sort_indices()
    coo = Xs.tocoo()
    rows, cols, x = coo.row, coo.col, coo.data
    W, H = W.astype(np.float64).copy(), H.astype(np.float64).copy()
    for _ in range(iters):
        q = x / np.einsum("ij,ij->i", W[rows], H[:, cols].T)
        Q = sp.csr_matrix((q, (rows, cols)), shape=Xs.shape)
        H *= (Q.T @ W).T / W.sum(axis=0)[:, None]
        q = x / np.einsum("ij,ij->i", W[rows], H[:, cols].T)
        Q = sp.csr_matrix((q, (rows, cols)), shape=Xs.shape)
        W *= (Q @ H.T) / H.sum(axis=1)[None, :]
    p = np.einsum("ij,ij->i", W[rows], H[:, cols].T)
    obj = float(np.sqrt(np.sum((x - p) ** 2 - p ** 2) + np.sum((W.T @ W) * (H @ H.T))))
    return W, H, obj


def test_gather_form_reference_is_the_oracle(oracle):
    """The scipy restatement used at full size below IS the oracle's arithmetic (dense Float64, zeros -> lambda)."""
    import scipy.sparse as sp

    n, m, k = 300, 96, 7
    pos = oracle.uniform_fill(5, 0, n * m).reshape(n, m) < 0.05
    X = np.where(pos, 1 + 4 * oracle.uniform_fill(6, 0, n * m).reshape(n, m), 0.0)
    X[np.arange(n), np.arange(n) % m] = 0.5
    W0, H0 = oracle.init_factors(77, n, m, k)
    ref = oracle.singlerun(X.astype(np.float32), k, W0, H0, maxiter=6, **NOSTOP)
    W, H, obj = _gather_form_mu(sp.csc_matrix(X.astype(np.float32)), W0, H0, 6)
    assert _rel(W @ H, ref["W"] @ ref["H"], X) <= 1e-12
    assert abs(obj - ref["objvalue"]) <= 1e-10 * ref["objvalue"]


def test_sparse_cfg4_full_size(NMFk, ctx, oracle):
    """100000 x 4096, 0.5 % fill (2.04 M non-zeros, rows of ~20 and columns of ~500 non-zeros, some rows EMPTY), one
    restart of ranks of every lane class of the gather kernels (k = 3, 8, 13, 20, 32, 40) for 6 iterations, against
    the Float64 gather-form reference from identical initial factors: W*H on 3000 rows, objective, and the C-ABI objective
    entry point on the result."""
    import scipy.sparse as sp

    n, m, fill = 100000, 4096, 0.005
    rng = np.random.default_rng(3)
    nnz = int(n * m * fill)
    Xs = sp.csc_matrix((rng.uniform(1, 5, nnz).astype(np.float32), (rng.integers(0, n, nnz), rng.integers(0, m, nnz))), shape=(n, m))
    Xs.sum_duplicates()
    lil = Xs.tolil()
    lil[12345, :] = 0  # an empty row and an empty column
    lil[:, 777] = 0
    Xs = sp.csc_matrix(lil)
    Xs.eliminate_zeros()
    ctx.set_X_sparse(Xs)
    assert ctx.nnz == Xs.nnz
    ks, iters = [3, 8, 13, 20, 32, 40], 6
    seeds = np.array([[NMFk.run_seed(4, k, 0)] for k in ks], dtype=np.uint64)
    res = ctx.mu_sweep(ks, 1, seeds=seeds, maxiter=iters, **NOSTOP)
    for q, k in enumerate(ks):
        W0, H0 = oracle.init_factors(int(seeds[q, 0]), n, m, k)
        W, H, obj = _gather_form_mu(Xs, W0, H0, iters)
        Wg, Hg = res[k]["W"][0].astype(np.float64), res[k]["H"][0].astype(np.float64)
        assert res[k]["iters"][0] == iters
        # (the sweep returns the factors with the reference's final scaling, so products are compared: 3000 rows of W*H)
        sel = np.r_[0:1000, 12000:13000, n - 1000:n]
        P, Pg = W[sel] @ H, Wg[sel] @ Hg
        assert np.linalg.norm(Pg - P) <= 2e-5 * np.linalg.norm(P), k
        assert abs(res[k]["objvalue"][0] - obj) <= 1e-5 * obj, k
        assert abs(ctx.frobenius(res[k]["W"][0], res[k]["H"][0]) - res[k]["objvalue"][0]) <= 1e-5 * obj
        assert (Wg[12345] <= 1e-30).all() and (Hg[:, 777] <= 1e-30).all()  # no data: the factors' rows go to zero
