"""CPU oracle for the NMFk.jl `execute(X, krange, nNMF; method=:simple)` path.

TEST INFRASTRUCTURE ONLY (see the header of nmfk_oracle.c): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the product package.

Two layers:
  * ctypes bindings to libnmfk_oracle.so (the C restatement of the numeric kernels), and
  * the orchestration of the reference restated in Python: `execute_run` (src/NMFkExecute.jl:483-711),
    `execute` (Exec:178-233, 236-329), `getk` (src/NMFkPostprocess.jl:7-41), `signalorder` (Post:148-158),
    plus small numpy twins of the C kernels used to cross-check the C code (tests/test_oracle_units.py).

All citations are relative to /root/reference.  Matrices are numpy arrays in Fortran (column-major) order.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STOP_MAXITER, STOP_STAGNATION, STOP_TOL, STOP_CONSISTENCY = 1, 2, 3, 4


class Params(C.Structure):
    _fields_ = [("tol", C.c_double), ("tolOF", C.c_double), ("lambda_", C.c_double), ("weight", C.c_double),
                ("maxiter", C.c_int64), ("maxreattempts", C.c_int32), ("maxbaditers", C.c_int32),
                ("stopconv", C.c_int32), ("Wfixed", C.c_int32), ("Hfixed", C.c_int32), ("tbits", C.c_int32),
                ("nthreads", C.c_int32)]


def make_params(tol=1e-19, tolOF=1e-3, lambda_=1e-32, weight=1.0, maxiter=10000, maxreattempts=2, maxbaditers=10,
                stopconv=1000, Wfixed=False, Hfixed=False, tbits=64, nthreads=1):
    """Defaults: Exec:729 (maxiter, tol) and Mult:24 (the rest).  nthreads: OpenMP threads of the C loops (results do
    not depend on it); 1 by default because a thread team per tiny loop is ruinous on many-core hosts."""
    return Params(tol, tolOF, lambda_, float(weight), int(maxiter), maxreattempts, maxbaditers, stopconv, int(Wfixed),
                  int(Hfixed), tbits, nthreads)


def build(force=False):
    so = os.path.join(_HERE, "libnmfk_oracle.so")
    src = os.path.join(_HERE, "nmfk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnmfk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libnmfk_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.nmfk_or_init.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, dp, dp]
        L.nmfk_or_uniform_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, dp]
        L.nmfk_or_preprocess.argtypes = [dp, C.c_int64, C.c_int64, C.c_double, bp, bp]
        L.nmfk_or_multiplicative.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, C.POINTER(Params), dp, dp, dp,
                                             C.POINTER(C.c_int64), ip, ip, dp]
        L.nmfk_or_frobenius.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, dp, dp]
        L.nmfk_or_frobenius.restype = C.c_double
        L.nmfk_or_singlerun.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, C.POINTER(Params), C.c_int32, dp, dp, dp,
                                        dp, C.POINTER(C.c_int64), ip, dp, dp]
        for suf, ct in (("f32", C.c_float), ("f64", C.c_double)):
            tp = C.POINTER(ct)
            getattr(L, "nmfk_or_clustersolutions_" + suf).argtypes = [tp, C.c_int64, C.c_int64, C.c_int64, ip, tp]
            getattr(L, "nmfk_or_finalize_" + suf).argtypes = [tp, ip, C.c_int64, C.c_int64, C.c_int64, tp, tp, tp]
            getattr(L, "nmfk_or_finalize_" + suf).restype = None
        L.nmfk_or_cluster_stats.argtypes = [dp, dp, ip, C.c_int64, C.c_int64, C.c_int64, C.c_int64, dp, dp, dp, dp]
        L.nmfk_or_cluster_stats.restype = None
        for suf, ct in (("f32", C.c_float), ("f64", C.c_double)):
            tp = C.POINTER(ct)
            getattr(L, "nmfk_or_kmeans_" + suf).argtypes = [tp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
                                                            ip, tp, tp, ip, dp, ip]
            getattr(L, "nmfk_or_robustkmeans_" + suf).argtypes = [tp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                                  C.c_double, C.c_uint64, ip, tp, tp, ip, dp, ip, ip, dp]
            getattr(L, "nmfk_or_point_silhouettes_" + suf).argtypes = [tp, C.c_int, C.c_int, ip, C.c_int, tp]
            getattr(L, "nmfk_or_point_silhouettes_" + suf).restype = None
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.array(a, dtype=np.float64, order="F", copy=True)


def tbits_of(X):
    return 32 if np.asarray(X).dtype == np.float32 else 64


def init_factors(seed, n, m, k):
    """Portable stand-in for `W = rand(n,k); H = rand(k,m)` (Mult:38,48): W first, then H."""
    W = np.empty((n, k), dtype=np.float64, order="F")
    H = np.empty((k, m), dtype=np.float64, order="F")
    lib().nmfk_or_init(C.c_uint64(seed), n, m, k, _dp(W), _dp(H))
    return W, H


def uniform_fill(seed, offset, count):
    out = np.empty(count, dtype=np.float64)
    lib().nmfk_or_uniform_fill(C.c_uint64(seed), C.c_uint64(offset), count, _dp(out))
    return out


def multiplicative(X, k, Winit, Hinit, params=None, trace=False, **kw):
    """NMFmultiplicative (Mult:24-127).  Returns dict(W,H,sse,iters,reason[,trace]).  Raises ValueError with the
    reference's message for negative entries (Mult:4-7)."""
    X = np.asarray(X)
    P = params or make_params(tbits=tbits_of(X), **kw)
    n, m = X.shape
    Xd, W, H = _f64(X), _f64(Winit), _f64(Hinit)
    assert W.shape == (n, k) and H.shape == (k, m)
    sse, iters, reason, nchecks = C.c_double(), C.c_int64(), C.c_int32(), C.c_int32()
    tr = np.zeros(max(int(P.maxiter) // 10, 1) if trace else 1, dtype=np.float64)
    rc = lib().nmfk_or_multiplicative(_dp(Xd), n, m, k, C.byref(P), _dp(W), _dp(H), C.byref(sse), C.byref(iters),
                                      C.byref(reason), C.byref(nchecks), _dp(tr) if trace else None)
    if rc != 0:
        raise ValueError("All matrix entries must be nonnegative!")
    out = dict(W=W, H=H, sse=sse.value, iters=iters.value, reason=reason.value)
    if trace:
        out["trace"] = tr[:nchecks.value].copy()
    return out


def frobenius(X, W, H):
    """normnan(X - W*H)  (Help:226-228)."""
    X = _f64(X)
    n, m = X.shape
    return lib().nmfk_or_frobenius(_dp(X), n, m, W.shape[1], _dp(_f64(W)), _dp(_f64(H)))


def broadcast_weight(weight, n, m):
    """Julia broadcasting of `weight` against the n x m residual (Mult:74; shapes allowed by Exec:484)."""
    w = np.asarray(weight, dtype=np.float64)
    if w.ndim == 1:
        assert w.shape[0] == n
        w = w[:, None]
    return np.asfortranarray(np.broadcast_to(w, (n, m)))


def singlerun(X, k, Winit, Hinit, modifymatrices=True, params=None, weight_array=None, normalizevector=None, **kw):
    """execute_singlerun_compute, :simple branch (Exec:729-807): (W, H, objvalue) + diagnostics.
    weight_array: array-valued `weight` (Mult:74); normalizevector: Mult:27-31, 119-122."""
    X = np.asarray(X)
    P = params or make_params(tbits=tbits_of(X), **kw)
    n, m = X.shape
    Xd, W, H = _f64(X), _f64(Winit), _f64(Hinit)
    obj, sse, iters, reason = C.c_double(), C.c_double(), C.c_int64(), C.c_int32()
    wa = None if weight_array is None else broadcast_weight(weight_array, n, m)
    nv = None if normalizevector is None else np.ascontiguousarray(normalizevector, dtype=np.float64)
    rc = lib().nmfk_or_singlerun(_dp(Xd), n, m, k, C.byref(P), int(modifymatrices), _dp(W), _dp(H), C.byref(obj),
                                 C.byref(sse), C.byref(iters), C.byref(reason), None if wa is None else _dp(wa),
                                 None if nv is None else _dp(nv))
    if rc != 0:
        raise ValueError("All matrix entries must be nonnegative!")
    return dict(W=W, H=H, objvalue=obj.value, sse=sse.value, iters=iters.value, reason=reason.value)


def _T(tbits):
    return (np.float32, C.c_float, "f32") if tbits == 32 else (np.float64, C.c_double, "f64")


def clustersolutions(Hs, tbits=64):
    """clustersolutions(factors, false) (Clus:425-517).  Hs: list/array of R matrices k x m (sorted by the
    caller).  Returns (labels k x R int32 1-based, centroids k x m)."""
    npT, cT, suf = _T(tbits)
    R = len(Hs)
    k, m = np.asarray(Hs[0]).shape
    F = np.stack([np.asarray(h, dtype=npT).flatten(order="F") for h in Hs]).copy()
    labels = np.zeros((k, R), dtype=np.int32, order="F")
    cent = np.zeros((k, m), dtype=npT, order="F")
    getattr(lib(), "nmfk_or_clustersolutions_" + suf)(F.ctypes.data_as(C.POINTER(cT)), R, k, m,
                                                      labels.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      cent.ctypes.data_as(C.POINTER(cT)))
    return labels, cent


def finalize_silhouettes(Hs, labels, tbits=64):
    """Silhouette part of finalize(Wa, Ha, idx, false) (Fin:36-66).
    Returns (D kR x kR, point silhouettes k x R, cluster silhouettes k)."""
    npT, cT, suf = _T(tbits)
    R = len(Hs)
    k, m = np.asarray(Hs[0]).shape
    F = np.stack([np.asarray(h, dtype=npT).flatten(order="F") for h in Hs]).copy()
    lab = np.asfortranarray(labels, dtype=np.int32)
    D = np.zeros((k * R, k * R), dtype=npT, order="F")
    ps = np.zeros((k, R), dtype=npT, order="F")
    cs = np.zeros(k, dtype=npT)
    tp = C.POINTER(cT)
    getattr(lib(), "nmfk_or_finalize_" + suf)(F.ctypes.data_as(tp), lab.ctypes.data_as(C.POINTER(C.c_int32)), R, k, m,
                                              D.ctypes.data_as(tp), ps.ctypes.data_as(tp), cs.ctypes.data_as(tp))
    return D, ps, cs


def cluster_stats(Ws, Hs, labels):
    """Cluster means / corrected variances of W and H (Fin:64-77)."""
    R = len(Ws)
    n, k = np.asarray(Ws[0]).shape
    m = np.asarray(Hs[0]).shape[1]
    Wst = np.stack([_f64(w).flatten(order="F") for w in Ws]).copy()
    Hst = np.stack([_f64(h).flatten(order="F") for h in Hs]).copy()
    lab = np.asfortranarray(labels, dtype=np.int32)
    Wm, Wv = (np.zeros((n, k), order="F") for _ in range(2))
    Hm, Hv = (np.zeros((k, m), order="F") for _ in range(2))
    lib().nmfk_or_cluster_stats(_dp(Wst), _dp(Hst), lab.ctypes.data_as(C.POINTER(C.c_int32)), R, n, k, m, _dp(Wm),
                                _dp(Hm), _dp(Wv), _dp(Hv))
    return Wm, Hm, Wv, Hv


# ----------------------------------------------------------------------------------------------------------
# host-level restatements
# ----------------------------------------------------------------------------------------------------------
def zerostoepsilon(x):
    """Help:529-543: values < eps(T)^2 become eps(T)^2, on a copy."""
    x = np.array(x, copy=True)
    e = np.finfo(x.dtype).eps ** 2
    x[x < e] = e
    return x


def ssqrnan(x):
    """Help:222-224."""
    x = np.asarray(x, dtype=np.float64)
    return float(np.sum(x[~np.isnan(x)] ** 2))


def normnan(x):
    """Help:226-228."""
    return math.sqrt(ssqrnan(x))


def signalorder(W, H):
    """Post:148-158: sortperm(desc) of sum(W[:,i:i]*H[i:i,:]) = colsum(W)_i * rowsum(H)_i.  0-based."""
    s = np.asarray(W).sum(axis=0) * np.asarray(H).sum(axis=1)
    return np.argsort(-s, kind="stable")


def getk(nkrange, robustness, cutoff=0.5, strict=True):
    """Post:7-41.  `robustness` is either len(nkrange) long or indexed by k (1-based, Julia style: element k-1)."""
    nkrange = list(nkrange)
    r = np.asarray(robustness, dtype=np.float64)
    if len(r) != len(nkrange):
        r = r[[k - 1 for k in nkrange]]
    if np.all(np.isnan(r)):
        return 0
    if len(nkrange) == 1:
        if strict:
            return nkrange[-1] if r[-1] > cutoff else None
        return nkrange[-1]
    idx = [i for i, v in enumerate(r) if v > cutoff]
    if not idx:
        if strict:
            return None
        rr = np.where(np.isnan(r), -np.inf, r)
        return nkrange[int(np.argmax(rr))]
    return nkrange[idx[-1]]


def cosine_dist_np(a, b):
    """Distances.cosine_dist (published definition)."""
    a, b = np.asarray(a), np.asarray(b)
    return max(1 - float(a @ b) / (math.sqrt(float(a @ a)) * math.sqrt(float(b @ b))), 0.0)


def silhouettes_np(assign, D):
    """Clustering.silhouettes (published definition): numpy twin used to cross-check the C code."""
    assign = np.asarray(assign)
    n = len(assign)
    cl = np.unique(assign)
    s = np.zeros(n)
    for i in range(n):
        own = assign[i]
        cnt = np.sum(assign == own)
        if cnt == 1:
            continue
        a = D[i, assign == own].sum() / (cnt - 1)
        b = min(D[i, assign == c].mean() for c in cl if c != own)
        s[i] = 0.0 if a == b else (b - a) / max(a, b)
    return s


def clustersolutions_np(Hs):
    """Numpy twin of Clus:425-517 (no zero-column fix), for cross-checking the C code on small cases."""
    F = [np.array(h, dtype=np.float64).T.copy() for h in Hs]  # m x k, columns are signals
    R, k = len(F), F[0].shape[1]
    cent = F[0]  # aliased running sum (Clus:453-455)
    labels = np.zeros((k, R), dtype=np.int32)
    labels[:, 0] = np.arange(1, k + 1)
    for t in range(1, R):
        D = np.array([[cosine_dist_np(F[t][:, f], cent[:, c]) for c in range(k)] for f in range(k)])
        D[np.isnan(D)] = 0
        while D.min() < np.inf:
            q = int(np.argmin(D.flatten(order="F")))
            f, c = q % k, q // k
            labels[f, t] = c + 1
            D[f, :] = np.inf
            D[:, c] = np.inf
            cent[:, c] += F[t][:, f]
    return labels, (cent / R).T


def execute_run(X, nk, nNMF, inits, acceptratio=1, acceptfactor=math.inf, best=True, nanaction="zeroed", params=None,
                clusterWmatrix=False, weight_array=None, normalizevector=None, **kw):
    """execute_run (Exec:483-711), serial branch, clusterWmatrix=false, mixture=:null.

    inits: list of (Winit, Hinit) per run (the oracle is RNG-free; see init_factors).
    Returns dict with Wa, Ha, phi, minsilhouette, aic and the intermediates tests compare against."""
    X = np.asarray(X)
    tb = tbits_of(X)
    npT = np.float32 if tb == 32 else np.float64
    n, m = X.shape
    P = params or make_params(tbits=tb, **kw)
    modifymatrices = not (P.Wfixed or P.Hfixed)  # Exec:486-489
    WBig, HBig, objvalue, iters, reasons = [], [], [], [], []
    for i in range(nNMF):
        r = singlerun(X, nk, inits[i][0], inits[i][1], modifymatrices=modifymatrices, params=P,
                      weight_array=weight_array, normalizevector=normalizevector)
        WBig.append(r["W"].astype(npT))  # Exec:529-531: stored as Matrix{T}
        HBig.append(r["H"].astype(npT))
        objvalue.append(npT(r["objvalue"]))
        iters.append(r["iters"])
        reasons.append(r["reason"])
    objvalue = np.array(objvalue, dtype=npT)
    idxsort = np.argsort(objvalue, kind="stable")  # Exec:545 (NaN last, as Julia's isless)
    bestIdx = int(idxsort[0])
    Wbest, Hbest = WBig[bestIdx].copy(), HBig[bestIdx].copy()
    idxrat = np.ones(nNMF, dtype=bool)
    if acceptratio < 1:  # Exec:552-558: keeps the first ceil(R*ratio) POSITIONS
        ccc = int(math.ceil(nNMF * acceptratio))
        idxrat = np.array([True] * ccc + [False] * (nNMF - ccc))
    idxcut = np.ones(nNMF, dtype=bool)
    if acceptfactor < math.inf:  # Exec:559-565
        idxcut = objvalue[idxsort] < objvalue[bestIdx] * acceptfactor
    idxnan = np.ones(nNMF, dtype=bool)
    if nanaction == "zeroed":  # Exec:567-580
        for i in idxsort:
            WBig[i][np.isnan(WBig[i])] = 0
            HBig[i][np.isnan(HBig[i])] = 0
    elif nanaction == "removed":  # Exec:581-595
        for i in idxsort:
            if np.isnan(WBig[i]).any() or np.isnan(HBig[i]).any():
                idxnan[i] = False
    idxsol = idxrat & idxcut & idxnan  # Exec:596 (indexes the SORTED list)
    Ws = [WBig[i] for i in idxsort[idxsol]]
    Hs = [HBig[i] for i in idxsort[idxsol]]
    out = dict(objvalue=objvalue, idxsort=idxsort, iters=iters, reasons=reasons, WBig=WBig, HBig=HBig)
    minsil = 1.0
    if nk > 1:
        if clusterWmatrix:  # Exec:621; in-place mutation of the first solution's W (Clus:453-455, 484, 512)
            labels, cent = clustersolutions([w.T for w in Ws], tbits=tb)
            Ws[0][...] = cent.T
        else:
            labels, cent = clustersolutions(Hs, tbits=tb)  # Exec:623
        ci = labels[:, 0]
        Wb0, Hb0 = WBig[bestIdx], HBig[bestIdx]
        for i, c in enumerate(ci):  # Exec:631-635
            Wbest[:, i] = Wb0[:, c - 1]
            Hbest[i, :] = Hb0[c - 1, :]
        D, psil, csil = finalize_silhouettes([w.T for w in Ws] if clusterWmatrix else Hs, labels, tbits=tb)  # Exec:637
        minsil = float(np.min(csil))  # Exec:638
        out.update(labels=labels, centroids=cent, psil=psil, csil=csil, D=D)
        if not best:
            Wm, Hm, Wv, Hv = cluster_stats(Ws, Hs, labels)
            out.update(Wmean=Wm.astype(npT), Hmean=Hm.astype(npT), Wvar=Wv.astype(npT), Hvar=Hv.astype(npT))
    if best:
        Wa, Ha = Wbest, Hbest  # Exec:655-658
    elif nk == 1:
        # Exec:648 -> Fin:114-118: finalize(WBig[idxsol], HBig[idxsol]) -- the mask built for the SORTED list is
        # applied to the unsorted vectors; the first survivor's W (n x 1) and H (1 x m), means over the unit dimension
        first = int(np.flatnonzero(idxsol)[0])
        Wa = WBig[first].mean(axis=1, keepdims=True)
        Ha = HBig[first].mean(axis=0, keepdims=True)
    else:
        Wa, Ha = out["Wmean"], out["Hmean"]
    E = np.asarray(X, dtype=np.float64) - np.asarray(Wa, dtype=npT) @ np.asarray(Ha, dtype=npT)  # Exec:664-667
    E[np.isnan(E)] = 0
    phi = float(npT(np.linalg.norm(E)))
    nobs = int(np.sum(~np.isnan(X)))  # Exec:697-708
    nparam = Wa.size + Ha.size
    aic = 2 * nparam + nobs * math.log(phi / nobs) if phi > 0 else -math.inf
    out.update(Wa=Wa, Ha=Ha, phi=phi, minsilhouette=minsil, aic=aic)
    return out


def execute(X, nkrange, nNMF=10, cutoff=0.5, seed=0, inits=None, **kw):
    """execute(X, nkrange, nNMF; method=:simple, load=false, save=false) (Exec:178-233 + 236-329).

    Returns (W, H, fitquality, robustness, aic, kopt, details): W/H dicts keyed by k; fitquality/robustness/aic
    arrays indexed by k-1 (Julia's 1-based vectors of length maxk, [1] = Inf / -1)."""
    X = np.asarray(X)
    npT = np.float32 if tbits_of(X) == 32 else np.float64
    n, m = X.shape
    nkrange = list(nkrange)
    maxk = max(nkrange)
    W, H, details = {}, {}, {}
    fit = np.zeros(maxk, dtype=npT)
    rob = np.zeros(maxk, dtype=npT)
    aic = np.zeros(maxk, dtype=npT)
    fit[0], rob[0] = np.inf, -1
    for nk in nkrange:
        ini = inits[nk] if inits is not None else [init_factors(run_seed(seed, nk, i), n, m, nk)
                                                   for i in range(nNMF)]
        r = execute_run(X, nk, nNMF, ini, **kw)
        so = signalorder(r["Wa"], r["Ha"])  # Exec:311-318
        W[nk], H[nk] = r["Wa"][:, so], r["Ha"][so, :]
        fit[nk - 1], rob[nk - 1], aic[nk - 1] = r["phi"], r["minsilhouette"], r["aic"]
        r["signalorder"] = so
        details[nk] = r
    if np.all(np.isinf(fit[[k - 1 for k in nkrange]])):  # Exec:206-208
        kopt = 0
    else:
        for nk in nkrange:  # Exec:211-222
            fit[nk - 1] = normnan(np.asarray(X, dtype=np.float64) - W[nk].astype(npT) @ H[nk].astype(npT))
        kopt = getk(nkrange, rob[[k - 1 for k in nkrange]], cutoff)
    return W, H, fit, rob, aic, kopt, details


def run_seed(seed, nk, run):
    """Seed of restart `run` (0-based) for rank nk: the same rule the product's host code uses, so oracle and
    GPU start from identical factors.  (Reference: seed = kwseed + i per run, Exec:536.)"""
    return (int(seed) * 1000003 + nk * 1009 + run + 1) & 0x7FFFFFFFFFFFFFFF


# ---------------------------------------------------------------------------------------------------------
# robustkmeans (Clus:138-246).  X: d x n, columns = samples (as in the reference).
# ---------------------------------------------------------------------------------------------------------
def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def kmeans(X, k, maxiter=1000, tol=1e-32, seed=0):
    """One Clustering.kmeans(X, k; distance=CosineDist()) run (restated, see nmfk_oracle.c).  0-based assignments."""
    X = np.asarray(X)
    npT, cT, suf = _T(tbits_of(X))
    Xf = np.asfortranarray(X, dtype=npT)
    d, n = Xf.shape
    assign, counts = np.zeros(n, np.int32), np.zeros(k, np.int32)
    centers, costs = np.zeros((d, k), npT, order="F"), np.zeros(n, npT)
    tc, conv = C.c_double(0), C.c_int32(0)
    tp = C.POINTER(cT)
    it = getattr(lib(), "nmfk_or_kmeans_" + suf)(Xf.ctypes.data_as(tp), d, n, k, maxiter, tol, C.c_uint64(seed), _ip(assign),
                                                 centers.ctypes.data_as(tp), costs.ctypes.data_as(tp), _ip(counts),
                                                 C.byref(tc), C.byref(conv))
    return dict(assignments=assign, centers=centers, costs=costs, counts=counts, totalcost=tc.value, iterations=it,
                converged=bool(conv.value))


def robustkmeans_k(X, k, repeats=1000, maxiter=1000, tol=1e-32, seed=0, compute_silhouettes_flag=False):
    """robustkmeans(X, k::Integer, repeats) (Clus:172-246 without the JLD cache): best of `repeats` by total cost,
    clusters relabelled by decreasing size.  Returns a dict (assignments 1-based) [, silhouettes of the best run]."""
    X = np.asarray(X)
    npT, cT, suf = _T(tbits_of(X))
    Xf = np.asfortranarray(X, dtype=npT)
    d, n = Xf.shape
    assign, counts = np.zeros(n, np.int32), np.zeros(k, np.int32)
    centers, costs = np.zeros((d, k), npT, order="F"), np.zeros(n, npT)
    allc = np.zeros(repeats, np.float64)
    tc, br, it = C.c_double(0), C.c_int32(0), C.c_int32(0)
    tp = C.POINTER(cT)
    kf = getattr(lib(), "nmfk_or_robustkmeans_" + suf)(Xf.ctypes.data_as(tp), d, n, k, repeats, maxiter, tol, C.c_uint64(seed),
                                                       _ip(assign), centers.ctypes.data_as(tp), costs.ctypes.data_as(tp),
                                                       _ip(counts), C.byref(tc), C.byref(br), C.byref(it), _dp(allc))
    res = dict(assignments=assign, centers=centers[:, :kf], costs=costs, counts=counts[:kf], totalcost=tc.value,
               iterations=it.value, best_repeat=br.value, all_costs=allc, nclusters=kf)
    if not compute_silhouettes_flag:
        return res
    if assign.max() > 1:  # Clus:211-218
        sil = np.zeros(n, npT)
        getattr(lib(), "nmfk_or_point_silhouettes_" + suf)(Xf.ctypes.data_as(tp), d, n, _ip(assign), k, sil.ctypes.data_as(tp))
    else:
        sil = np.zeros(n, npT)
    return res, sil


def robustkmeans(X, krange, repeats=1000, best_method="worst_cliff", **kw):
    """robustkmeans(X, krange, repeats) (Clus:138-170): the k after the largest drop of the worst point silhouette
    (:worst_cliff) or of the worst cluster-mean silhouette (:worst_cluster_cliff)."""
    X = np.asarray(X)
    krange = [int(k) for k in krange]
    if krange[0] >= X.shape[1]:
        return None
    res, worst, cworst = [], [], []
    for k in krange:
        if k >= X.shape[1]:  # Clus:149-152 (`continue` leaves undefined entries in the reference)
            res.append(None)
            worst.append(np.nan)
            cworst.append(np.nan)
            continue
        r, sil = robustkmeans_k(X, k, repeats, compute_silhouettes_flag=True, **kw)
        r["silhouettes"] = sil
        r["mean_silhouette"] = float(np.mean(sil))
        r["worst_silhouette"] = float(np.min(sil))
        a = r["assignments"]
        first = list(dict.fromkeys(a.tolist()))
        r["cluster_silhouettes"] = [float(np.mean(sil[a == j])) for j in first]
        res.append(r)
        worst.append(r["worst_silhouette"])
        cworst.append(min(r["cluster_silhouettes"]))
    v = worst if best_method == "worst_cliff" else cworst
    if best_method not in ("worst_cliff", "worst_cluster_cliff"):
        raise ValueError("Unknown method: best_method must be :worst_cliff or :worst_cluster_cliff")
    drops = [v[i] - v[i + 1] for i in range(len(krange) - 1)]
    ki = int(np.argmax(drops)) + 1  # findmax: first maximum
    return res[ki], krange[ki], res
