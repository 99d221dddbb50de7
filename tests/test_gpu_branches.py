"""execute_run's solution filters and the best=false return, end to end on the GPU against oracle.execute_run
(src/NMFkExecute.jl:552-596 acceptratio / acceptfactor / nanaction, 640-658 best=false)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NOSTOP = dict(maxbaditers=10 ** 9)


@pytest.fixture(scope="module")
def NMFk():
    import nmfk_jl_amd

    return nmfk_jl_amd


def _case(oracle, n=48, m=20, k0=3, seed=61):
    W0 = oracle.uniform_fill(seed, 0, n * k0).reshape(n, k0)
    H0 = oracle.uniform_fill(seed + 1, 0, k0 * m).reshape(k0, m)
    return (W0 @ H0 + 0.05 * oracle.uniform_fill(seed + 2, 0, n * m).reshape(n, m)).astype(np.float32)


def _inits(NMFk, oracle, seed, n, m, k, R):
    return [oracle.init_factors(NMFk.run_seed(seed, k, r), n, m, k) for r in range(R)]


def _compare(got, ref, k):
    Wa, Ha, phi, sil, aic, extra = got
    assert (extra["idxsort"] == ref["idxsort"]).all()
    assert abs(phi - ref["phi"]) <= 1e-3 * ref["phi"]
    assert abs(aic - ref["aic"]) <= 1e-3 * abs(ref["aic"]) + 1e-2
    if k > 1:
        assert (extra["labels"] == ref["labels"]).all()
        assert abs(sil - ref["minsilhouette"]) <= 1e-3
    np.testing.assert_allclose(Wa @ Ha, ref["Wa"] @ ref["Ha"], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("opts", [dict(acceptratio=0.5), dict(acceptfactor=1.5), dict(acceptratio=0.75, acceptfactor=1.2),
                                  dict(best=False), dict(best=False, acceptratio=0.5)])
def test_execute_run_filters_and_best_false(NMFk, oracle, opts):
    X = _case(oracle)
    n, m = X.shape
    k, R, iters = 3, 8, 150
    with pytest.warns(UserWarning) if ("acceptratio" in opts or "acceptfactor" in opts) else _nullcontext():
        got = NMFk.execute_run(X, k, R, seed=9, maxiter=iters, return_details=True, **NOSTOP, **opts)
    ref = oracle.execute_run(X, k, R, _inits(NMFk, oracle, 9, n, m, k, R), maxiter=iters, **NOSTOP, **opts)
    nsol = got[5]["labels"].shape[1]
    assert nsol == ref["labels"].shape[1]
    if "acceptratio" in opts and "acceptfactor" not in opts:
        assert nsol == math.ceil(R * opts["acceptratio"])
    _compare(got, ref, k)
    if not opts.get("best", True):
        np.testing.assert_allclose(got[0], ref["Wmean"], rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(got[5]["Wvar"], ref["Wvar"], rtol=5e-3, atol=1e-6)


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_best_false_single_signal_takes_first_restart(NMFk, oracle):
    """nk = 1, best=false: finalize(WBig[idxsol], HBig[idxsol]) (Exec:648, Fin:114-118) returns the FIRST restart in
    restart order, not the one with the lowest objective."""
    X = _case(oracle, k0=2, seed=71)
    n, m = X.shape
    R = 5
    got = NMFk.execute_run(X, 1, R, seed=4, maxiter=40, best=False, return_details=True, **NOSTOP)
    ref = oracle.execute_run(X, 1, R, _inits(NMFk, oracle, 4, n, m, 1, R), maxiter=40, best=False, **NOSTOP)
    assert got[3] == 1 and ref["minsilhouette"] == 1
    np.testing.assert_allclose(got[0], ref["Wa"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got[1], ref["Ha"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got[0], ref["WBig"][0], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("nanaction", ["removed", "zeroed"])
def test_nanaction_with_a_nan_restart(NMFk, oracle, nanaction):
    """A restart whose initial W has an all-zero column ends in NaNs (0 * x / colsum(W) = 0 / 0 in Mult:67, and Julia's
    max.(H, eps()) at Mult:99 keeps a NaN); Exec:567-595 zeroes such a solution or drops one.  The product's
    post-processing gets the GPU restarts, the oracle its own from the same initial factors."""
    from nmfk_jl_amd.execute import _execute_run_post

    X = _case(oracle, seed=81)
    n, m = X.shape
    k, R, iters = 3, 6, 60
    inits = _inits(NMFk, oracle, 12, n, m, k, R)
    inits[2][0][:, 1] = 0.0
    ctx = NMFk.Context(0)
    ctx.set_X(X)
    res = ctx.mu_sweep([k], R, Winit={k: np.stack([w for w, _ in inits])}, Hinit={k: np.stack([h for _, h in inits])},
                       maxiter=iters, **NOSTOP)[k]
    assert np.isnan(res["H"][2]).any() and not np.isnan(res["H"][[0, 1, 3, 4, 5]]).any()
    with pytest.warns(UserWarning) if nanaction == "zeroed" else _nullcontext():
        got = _execute_run_post(ctx, X, k, R, res, nanaction=nanaction)
    ref = oracle.execute_run(X, k, R, inits, maxiter=iters, nanaction=nanaction, **NOSTOP)
    # objvalue = normnan(X - W*H) (Exec:791-792) skips the NaN residuals: an all-NaN restart scores 0 and sorts FIRST
    assert res["objvalue"][2] == 0 and ref["objvalue"][2] == 0
    assert (got[5]["idxsort"] == ref["idxsort"]).all() and int(got[5]["idxsort"][0]) == 2
    # Exec:581-595 marks idxnan[i] with the UNSORTED index i and Exec:596 combines it with masks over sorted positions:
    # the product keeps that literal behaviour, so the same number of solutions survives on both sides
    assert got[5]["labels"].shape[1] == ref["labels"].shape[1] == (R - 1 if nanaction == "removed" else R)
    assert abs(got[2] - ref["phi"]) <= 1e-3 * max(ref["phi"], 1e-6)
    ctx.close()
