"""Cross-checks of the C oracle against independent implementations available in this image
(numpy twins, scipy, scikit-learn) for the arithmetic the reference delegates to Distances.jl / Clustering.jl."""
import numpy as np
import pytest


def _random_solutions(rng, R, k, m):
    base = rng.random((k, m)) ** 3
    Hs = []
    for _ in range(R):
        perm = rng.permutation(k)
        Hs.append(np.asfortranarray(base[perm] * (1 + 0.05 * rng.random((k, m)))))
    return Hs


@pytest.mark.parametrize("tb", [32, 64])
def test_clustersolutions_c_vs_numpy(oracle, tb):
    rng = np.random.default_rng(0)
    for (R, k, m) in [(2, 2, 5), (6, 3, 7), (10, 5, 12), (4, 8, 33)]:
        Hs = _random_solutions(rng, R, k, m)
        lab_c, cen_c = oracle.clustersolutions(Hs, tbits=tb)
        lab_n, cen_n = oracle.clustersolutions_np(Hs)
        assert (lab_c == lab_n).all()
        np.testing.assert_allclose(cen_c, cen_n, rtol=2e-6 if tb == 32 else 1e-12)
        for t in range(R):
            assert sorted(lab_c[:, t].tolist()) == list(range(1, k + 1))


def test_clustersolutions_zero_column_fix(oracle):
    # Clus:436-450: an all-zero signal triggers the bias row; labels stay permutations, no NaNs
    rng = np.random.default_rng(1)
    Hs = _random_solutions(rng, 4, 3, 6)
    Hs[2][1, :] = 0
    lab, cen = oracle.clustersolutions(Hs, tbits=64)
    for t in range(4):
        assert sorted(lab[:, t].tolist()) == [1, 2, 3]
    assert np.isfinite(cen).all()


@pytest.mark.parametrize("tb", [32, 64])
def test_finalize_vs_scipy_sklearn(oracle, tb):
    from scipy.spatial.distance import cdist
    from sklearn.metrics import silhouette_samples

    rng = np.random.default_rng(2)
    for (R, k, m) in [(5, 3, 9), (10, 4, 20), (8, 2, 6)]:
        Hs = _random_solutions(rng, R, k, m)
        lab, _ = oracle.clustersolutions(Hs, tbits=tb)
        D, ps, cs = oracle.finalize_silhouettes(Hs, lab, tbits=tb)
        Z = np.vstack([np.asarray(h, dtype=np.float64) for h in Hs])  # vcat(Ha...): rows a + r*k
        Dref = np.maximum(cdist(Z, Z, "cosine"), 0)
        np.fill_diagonal(Dref, 0)
        tol = 5e-6 if tb == 32 else 1e-12
        np.testing.assert_allclose(D, Dref, atol=tol)
        assign = lab.flatten(order="F")
        sref = silhouette_samples(np.asarray(D, dtype=np.float64), assign, metric="precomputed")
        np.testing.assert_allclose(ps.flatten(order="F"), sref, atol=20 * tol)
        np.testing.assert_allclose(ps.flatten(order="F"), oracle.silhouettes_np(assign, np.asarray(D, np.float64)),
                                   atol=20 * tol)
        for c in range(k):
            np.testing.assert_allclose(cs[c], sref[assign == c + 1].mean(), atol=20 * tol)


def test_cluster_stats_vs_numpy(oracle):
    rng = np.random.default_rng(3)
    R, n, k, m = 6, 7, 3, 5
    Hs = _random_solutions(rng, R, k, m)
    Ws = [np.asfortranarray(rng.random((n, k))) for _ in range(R)]
    lab, _ = oracle.clustersolutions(Hs)
    Wm, Hm, Wv, Hv = oracle.cluster_stats(Ws, Hs, lab)
    for c in range(k):
        hs = np.stack([Hs[r][list(lab[:, r]).index(c + 1), :] for r in range(R)])
        ws = np.stack([Ws[r][:, list(lab[:, r]).index(c + 1)] for r in range(R)])
        np.testing.assert_allclose(Hm[c], hs.mean(0))
        np.testing.assert_allclose(Hv[c], hs.var(0, ddof=1))
        np.testing.assert_allclose(Wm[:, c], ws.mean(0))
        np.testing.assert_allclose(Wv[:, c], ws.var(0, ddof=1))


def _mu_numpy(X, W, H, iters):
    """Literal numpy transcription of Mult:67,70 (no NaNs), used to check the fused C loops."""
    for _ in range(iters):
        H = H * (W.T @ (X / (W @ H))) / W.sum(axis=0)[:, None]
        W = W * ((X / (W @ H)) @ H.T) / H.sum(axis=1)[None, :]
    return W, H


def test_mu_updates_vs_numpy(oracle):
    rng = np.random.default_rng(4)
    n, m, k = 23, 11, 3
    X = rng.random((n, m))
    W0, H0 = oracle.init_factors(9, n, m, k)
    r = oracle.multiplicative(X, k, W0, H0, maxiter=9)  # no check block before iteration 10
    Wn, Hn = _mu_numpy(X, W0.copy(), H0.copy(), 9)
    np.testing.assert_allclose(r["W"], Wn, rtol=1e-11)
    np.testing.assert_allclose(r["H"], Hn, rtol=1e-11)
    assert r["iters"] == 9 and r["reason"] == oracle.STOP_MAXITER
    np.testing.assert_allclose(r["sse"], ((X - Wn @ Hn) ** 2).sum(), rtol=1e-11)


def test_mu_nan_imputation_vs_numpy(oracle):
    """Mult:17-20,72: NaN -> lambda on entry, then EM imputation with the post-update W*H every iteration."""
    rng = np.random.default_rng(5)
    n, m, k = 12, 9, 2
    X = rng.random((n, m))
    mask = rng.random((n, m)) < 0.2
    X[mask] = np.nan
    X[0, 0] = 0.0
    W0, H0 = oracle.init_factors(3, n, m, k)
    r = oracle.multiplicative(X, k, W0, H0, maxiter=7)
    Xw = X.copy()
    Xw[Xw <= 0] = 1e-32
    Xw[mask] = 1e-32
    W, H = W0.copy(), H0.copy()
    for _ in range(7):
        H = H * (W.T @ (Xw / (W @ H))) / W.sum(axis=0)[:, None]
        W = W * ((Xw / (W @ H)) @ H.T) / H.sum(axis=1)[None, :]
        Xw[mask] = (W @ H)[mask]
    np.testing.assert_allclose(r["W"], W, rtol=1e-10)
    np.testing.assert_allclose(r["H"], H, rtol=1e-10)
    E = (X - W @ H)[~mask]
    np.testing.assert_allclose(r["sse"], (E ** 2).sum(), rtol=1e-10)


def test_mu_stop_rule_state_machine(oracle):
    """Replays Mult:73-98 on the recorded objective trace and checks iters / stop reason."""
    rng = np.random.default_rng(6)
    X = rng.random((40, 12))
    W0, H0 = oracle.init_factors(1, 40, 12, 3)
    r = oracle.multiplicative(X, 3, W0, H0, trace=True)
    best, bad, re_, it = np.inf, 0, 0, 0
    for obj in r["trace"]:
        it += 10
        if obj < best:
            bad = bad + 1 if best - obj < 1e-3 else 0
            best = obj
        else:
            bad += 1
        if bad >= 10:
            re_ += 1
            bad = 0
        if re_ >= 2:
            break
    assert it == r["iters"] and r["reason"] == oracle.STOP_STAGNATION
    assert len(r["trace"]) == r["iters"] // 10


def test_mu_fixed_factors(oracle):
    rng = np.random.default_rng(7)
    X = rng.random((10, 8))
    W0, H0 = oracle.init_factors(2, 10, 8, 2)
    r = oracle.multiplicative(X, 2, W0, H0, maxiter=30, Hfixed=True)
    eps = 2.220446049250313e-16
    np.testing.assert_array_equal(r["H"], np.maximum(H0, eps))
    r = oracle.multiplicative(X, 2, W0, H0, maxiter=30, Wfixed=True)
    np.testing.assert_array_equal(r["W"], np.maximum(W0, eps))


def test_threads_do_not_change_results(oracle):
    rng = np.random.default_rng(8)
    X = rng.random((64, 32)).astype(np.float32)
    W0, H0 = oracle.init_factors(4, 64, 32, 4)
    a = oracle.multiplicative(X, 4, W0, H0, maxiter=50, nthreads=1)
    b = oracle.multiplicative(X, 4, W0, H0, maxiter=50, nthreads=4)
    assert (a["W"] == b["W"]).all() and (a["H"] == b["H"]).all() and a["sse"] == b["sse"]


def test_rng_properties(oracle):
    u = oracle.uniform_fill(42, 0, 100000)
    assert u.min() > 0 and u.max() < 1
    assert (u.astype(np.float32).astype(np.float64) == u).all()  # exactly representable in fp32
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    assert (oracle.uniform_fill(42, 10, 5) == u[10:15]).all()
    assert (oracle.uniform_fill(43, 0, 5) != u[:5]).all()


# ---------------------------------------------------------------------------------------------------------
# robustkmeans (Clus:138-246)
# ---------------------------------------------------------------------------------------------------------
def _direction_clusters(oracle, d, per, nc, seed, noise=0.05):
    """nc groups of `per` samples, each along its own direction (cosine distance separates them), random lengths."""
    u = oracle.uniform_fill(seed, 0, d * nc + 2 * per * nc + d * per * nc)
    dirs = 0.1 + np.eye(d)[:, :nc] if nc <= d else u[:d * nc].reshape(d, nc)
    cols = []
    for c in range(nc):
        length = 0.5 + 1.5 * u[d * nc + c * per:d * nc + (c + 1) * per]
        nz = u[d * nc + 2 * per * nc + c * d * per:d * nc + 2 * per * nc + (c + 1) * d * per].reshape(d, per)
        cols.append(dirs[:, [c]] * length[None, :] + noise * nz)
    X = np.concatenate(cols, axis=1)
    perm = np.argsort(oracle.uniform_fill(seed + 1, 0, X.shape[1]))  # shuffle the columns
    return np.asfortranarray(X[:, perm])


def test_kmeans_reference_test_vector(oracle):
    """test/test_cluster_unit.jl:6-18: valid assignments on the 2 x 4 example (k = 2, 5 repeats, maxiter 50, tol 1e-8)."""
    X = np.array([[1.0, 1.1, 10.0, 10.1], [1.0, 0.9, 10.0, 9.9]])
    r = oracle.robustkmeans_k(X, 2, 5, maxiter=50, tol=1e-8)
    assert len(r["assignments"]) == 4 and sorted(set(r["assignments"].tolist())) == [1, 2]
    assert r["centers"].shape == (2, 2)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_kmeans_invariants(oracle, T):
    X = _direction_clusters(oracle, 5, 30, 3, seed=7).astype(T)
    r = oracle.kmeans(X, 3, seed=11)
    a = r["assignments"]
    assert r["converged"] and r["iterations"] >= 1 and (np.bincount(a, minlength=3) == r["counts"]).all()
    for c in range(3):  # centres = mean of the members; costs = cosine distance to the own centre
        np.testing.assert_allclose(r["centers"][:, c], X[:, a == c].mean(axis=1), rtol=2e-6)
    own = np.array([oracle.cosine_dist_np(r["centers"][:, a[j]].astype(np.float64), X[:, j].astype(np.float64)) for j in range(X.shape[1])])
    np.testing.assert_allclose(r["costs"], own, atol=2e-6)
    assert abs(r["totalcost"] - r["costs"].astype(np.float64).sum()) < 1e-9
    other = np.array([[oracle.cosine_dist_np(r["centers"][:, c].astype(np.float64), X[:, j].astype(np.float64)) for c in range(3)]
                      for j in range(X.shape[1])])
    assert (other.min(axis=1) >= own - 1e-6).all()  # every sample sits in its nearest cluster


def test_robustkmeans_finds_planted_clusters_and_sorts_by_size(oracle):
    X = np.concatenate([_direction_clusters(oracle, 4, 25, 3, seed=3), _direction_clusters(oracle, 4, 10, 1, seed=5)], axis=1)
    r, sil = oracle.robustkmeans_k(X, 3, 30, compute_silhouettes_flag=True)
    assert r["nclusters"] == 3 and (np.diff(r["counts"]) <= 0).all() and r["counts"].sum() == X.shape[1]
    assert (np.bincount(r["assignments"])[1:] == r["counts"]).all()
    assert r["totalcost"] == r["all_costs"].min() and r["best_repeat"] == int(np.argmin(r["all_costs"]))
    Z = np.maximum(X, np.finfo(np.float64).eps ** 2)
    D = np.array([[oracle.cosine_dist_np(Z[:, i], Z[:, j]) for j in range(X.shape[1])] for i in range(X.shape[1])])
    np.testing.assert_allclose(sil, oracle.silhouettes_np(r["assignments"], D), atol=1e-12)
    assert sil.min() > 0.5


def test_robustkmeans_krange_selection(oracle):
    X = _direction_clusters(oracle, 5, 20, 3, seed=9)
    best, kbest, allr = oracle.robustkmeans(X, [2, 3, 4, 5], 20)
    worst = [r["worst_silhouette"] for r in allr]
    assert kbest == [2, 3, 4, 5][int(np.argmax([worst[i] - worst[i + 1] for i in range(3)])) + 1]  # Clus:160
    assert worst[1] > 0.8 > worst[2]  # three planted directions: the cliff is between k = 3 and k = 4
    assert oracle.robustkmeans(X[:, :2], [2, 3], 5) is None  # Clus:139-142
