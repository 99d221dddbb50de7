"""Host side of `NMFk.robustkmeans` (src/NMFkCluster.jl:138-246): the repeats run on the GPU
(nmfk_robustkmeans), the selection over a range of k and the result cache stay on the host.

Differences from the reference, all stated: the random draws come from the library's counter-based generator (Julia's
stream is not reproducible), the silhouettes are only computed for the winning repeat (the reference computes them for
every repeat and keeps the winner's: same result).  The cache is the reference's file: `<case>-<k>-<d>_<n>-<repeats>.jld` with
"assignments" = the winning Clustering.KmeansResult (sorted) and "best_silhouettes" (Clus:173-199, 236-244), written and
read by jldfile.py in the layout of the Julia-written fixtures under tests/golden/."""
import os
import warnings

import numpy as np

from . import _lib, resultio


def _context(ctx, device):
    if ctx is not None:
        return ctx
    from .execute import _context as ec

    return ec(device)


def robustkmeans(X, krange, repeats=1000, *, best_method="worst_cliff", maxiter=1000, tol=1e-32, resultdir=".",
                 casefilename="assignments", load=False, save=False, compute_silhouettes_flag=False, seed=0, ctx=None,
                 device=None):
    """robustkmeans(X, k::Integer, repeats) -> result dict [, silhouettes]   (Clus:172-246)
    robustkmeans(X, krange, repeats) -> result dict of the selected k            (Clus:138-170), None when
    krange[1] >= size(X, 2).  X: d x n, columns = samples; assignments are 1-based and sorted by cluster size."""
    X = np.asarray(X)
    if isinstance(krange, (int, np.integer)):
        return _robust_k(X, int(krange), int(repeats), maxiter, tol, resultdir, casefilename, load, save,
                         compute_silhouettes_flag, seed, _context(ctx, device))
    ks = [int(k) for k in krange]
    if best_method not in ("worst_cliff", "worst_cluster_cliff"):
        raise ValueError("Unknown method: best_method must be :worst_cliff or :worst_cluster_cliff")
    if ks[0] >= X.shape[1]:  # Clus:139-142
        return None
    c = _context(ctx, device)
    res, worst, cworst = [], [], []
    for k in ks:
        if k >= X.shape[1]:  # Clus:149-152
            res.append(None)
            worst.append(np.nan)
            cworst.append(np.nan)
            continue
        r, sil = _robust_k(X, k, int(repeats), maxiter, tol, resultdir, casefilename, load, save, True, seed, c)
        a = r["assignments"]
        first = list(dict.fromkeys(a.tolist()))
        r.update(silhouettes=sil, mean_silhouette=float(np.mean(sil)), worst_silhouette=float(np.min(sil)),
                 cluster_silhouettes=[float(np.mean(sil[a == j])) for j in first])
        res.append(r)
        worst.append(r["worst_silhouette"])
        cworst.append(min(r["cluster_silhouettes"]))
    v = worst if best_method == "worst_cliff" else cworst
    drops = [v[i] - v[i + 1] for i in range(len(ks) - 1)]
    ki = int(np.argmax(drops)) + 1  # Clus:160-162: findmax(...)[2] + 1
    out = res[ki]
    out["k"] = ks[ki]
    return out


def _robust_k(X, k, repeats, maxiter, tol, resultdir, casefilename, load, save, sil_flag, seed, ctx):
    fn = os.path.join(resultdir, f"{casefilename}-{k}-{'_'.join(str(v) for v in X.shape)}-{repeats}.jld")  # Clus:174,237
    if load and casefilename != "":
        if os.path.isfile(fn):  # Clus:175-196
            try:
                f = resultio.load(fn)
            except (ValueError, AssertionError, KeyError, OSError) as e:  # unreadable file: recompute (and rewrite it)
                warnings.warn(f"'{fn}' cannot be read ({e})")
                f = {}
            sc = f.get("assignments")
            if isinstance(sc, dict) and "assignments_" in sc and (not sil_flag or "best_silhouettes" in f):
                a = np.asarray(sc["assignments_"], dtype=np.int32)
                kf = int(len(np.unique(a)))
                ck = np.asarray(sc["centers_"], dtype=np.float32)
                nk = np.asarray(sc["counts_"], dtype=np.int32)
                # the file holds the KmeansResult (k columns / entries); the dict shows the clusters found, like the computed
                # path.  best_repeat / all_costs are diagnostics of a computation and are not part of the reference's file:
                # the key is absent (not None) after a cache load on BOTH paths' consumers (`res.get(...)`).
                res = dict(assignments=a, centers=ck[:, :kf], costs=np.asarray(sc["costs_"], dtype=np.float32), counts=nk[:kf],
                           totalcost=float(sc["totalcost_"]), iterations=int(sc["iterations_"]), converged=bool(sc["converged_"]),
                           nclusters=kf, centers_k=ck, counts_k=nk)
                return (res, np.asarray(f["best_silhouettes"], dtype=np.float32)) if sil_flag else res
            warnings.warn(f"Failed to load robust k-means results from '{fn}'; Robust k-means analysis will be executed ...")
    out = ctx.robustkmeans(X, k, repeats, maxiter=maxiter, tol=tol, seed=seed, compute_silhouettes_flag=sil_flag)
    res, sil = out if sil_flag else (out, None)
    if res["nclusters"] < k:  # Clus:232-234
        warnings.warn(f"Robust k-means analysis could not find {k} clusters! Only {res['nclusters']} clusters were found.")
    if save and casefilename != "":  # Clus:236-244: JLD.save(filename, "assignments", sc[, "best_silhouettes", ...])
        os.makedirs(resultdir, exist_ok=True)
        # Clustering.KmeansResult: centers d x k and counts / wcounts of length k whatever the clusters found (zero columns /
        # entries for the others), converged = the k-means convergence flag of the winning repeat
        sc = dict(centers_=np.asarray(res["centers_k"], np.float64), assignments_=np.asarray(res["assignments"], np.int64),
                  costs_=np.asarray(res["costs"], np.float64), counts_=np.asarray(res["counts_k"], np.int64),
                  wcounts_=np.asarray(res["counts_k"], np.int64), totalcost_=float(res["totalcost"]),
                  iterations_=int(res["iterations"]), converged_=int(bool(res["converged"])))
        payload = {"assignments": sc}
        if sil_flag:
            payload["best_silhouettes"] = np.asarray(sil, np.float64)
        resultio.save(fn, **payload)
    return (res, sil) if sil_flag else res


def sortclustering(c, rev=True):
    """sortclustering(c::AbstractVector) (Clus:248-262): labels renumbered by decreasing (rev) cluster size, ties in
    order of first appearance."""
    c = np.asarray(c)
    first = list(dict.fromkeys(c.tolist()))
    counts = [int(np.sum(c == a)) for a in first]
    order = sorted(range(len(first)), key=lambda i: -counts[i] if rev else counts[i])
    out = np.empty(len(c), dtype=np.int64)
    for new, i in enumerate(order):
        out[c == first[i]] = new + 1
    return out
