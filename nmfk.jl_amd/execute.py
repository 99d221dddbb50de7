"""Host-side mirror of NMFk.jl's `execute` for method=:simple (src/NMFkExecute.jl), driving libnmfk_hip.

    W, H, fitquality, robustness, aic, kopt = execute(X, range(2, 6), 10, save=False, load=False)

Same names, argument meaning, return shapes and error behaviour as the reference path:
  execute(X, nkrange, nNMF=10; ...)    Exec:178-233  -> (W, H, fitquality, robustness, aic, kopt)
  execute(X, nk::Integer, nNMF; ...)   Exec:236-329  -> (W, H, fit, robustness, aic)
  execute_run(X, nk, nNMF; ...)        Exec:483-711  -> (Wa, Ha, phi, minsilhouette, aic)
  getk / signalorder                   src/NMFkPostprocess.jl:7-41, 148-158
Julia's 1-based vectors of length maxk become Python sequences indexed by k-1 (W[k-1] is None where the
reference leaves #undef; fitquality[0] = Inf, robustness[0] = -1 as at Exec:200-201).

Everything numeric runs on the GPU through the C ABI (multiplicative updates, objective, normalisation,
clustering, silhouettes, fit re-checks); this file is orchestration only (sorting <= nNMF objective values,
the acceptance filters, k selection).  The reference's serial loops over k and over restarts become one
flat (k, restart) work list; with torch.distributed initialised the list is sharded by restart over the
ranks (parallel.py)."""
import dataclasses
import hashlib
import math
import os
import warnings

import numpy as np

from . import _lib, resultio
from ._lib import Context, NMFkError

_METHOD_ALIASES = {"multdiv", "multmse", "alspgrad"}  # Exec:138-147 -> method=:nmf (NMF.jl), not this path
_default_ctx = {}


def _context(device=None):
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _lib.device_count() > 1 else 0
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


def _is_sparse(X):
    return hasattr(X, "tocsc") and hasattr(X, "nnz")


@dataclasses.dataclass
class ExecuteOptions:
    """NMFk.ExecuteOptions (Exec:15-30): the keyword bundle of the options-based `execute` overloads (Exec:33-65)."""
    cutoff: float = 0.5
    clusterWmatrix: bool = False
    mixture: str = "null"
    method: str = "simple"
    algorithm: str = "multdiv"
    resultdir: str = "."
    load: bool = True
    save: bool = True
    casefilename: str = ""
    dims: object = (1, 2)
    loadonly: bool = False
    quiet: bool = False
    check_inputs: bool = True
    ordersignals: bool = True


_first_warning = True  # NMFk's module-level `first_warning` (Mult:8-15): the two warnings appear once per session


def _zero_line_warnings(X):
    """Mult:8-15: minimum(sum(X; dims=...)) == 0, once per session."""
    global _first_warning
    if not _first_warning:
        return
    _first_warning = False
    if _is_sparse(X):
        rs, cs = np.asarray(X.sum(axis=1)).ravel(), np.asarray(X.sum(axis=0)).ravel()
    else:
        Xa = np.asarray(X)
        rs, cs = Xa.sum(axis=1), Xa.sum(axis=0)
    # Julia's minimum() propagates NaN and `NaN == 0` is false: with a missing entry anywhere in X the reference stays
    # silent even when another row / column is all zero.  numpy's min() propagates NaN the same way.
    if rs.size and rs.min() == 0:
        warnings.warn("All matrix entries in a row should not be 0!")
    if cs.size and cs.min() == 0:
        warnings.warn("All matrix entries in a column should not be 0!")


def _upload(ctx, X, lambda_=1e-32):
    """Dense arrays go through NMFpreprocessing! (zeros -> lambda); scipy.sparse matrices select the gather kernels
    (zeros stay zeros, the same arithmetic to < 1e-30; BASELINE configs[3])."""
    if _is_sparse(X):
        ctx.set_X_sparse(X)
    else:
        ctx.set_X(X, lambda_)  # raises "All matrix entries must be nonnegative!" first (Mult:4-7)
    _zero_line_warnings(X)


def run_seed(seed, nk, run):
    """Seed of restart `run` (0-based) of rank nk.  The reference gives restart i the seed kwseed+i (Exec:536)
    and draws W = rand(n,k) then H = rand(k,m) from Julia's RNG (Mult:38,48); Julia's stream cannot be
    reproduced here, so a restart is identified by this seed in the library's own counter-based generator."""
    return (int(seed) * 1000003 + int(nk) * 1009 + int(run) + 1) & 0x7FFFFFFFFFFFFFFF


def signalorder(W, H):
    """Post:148-158: sortperm(rev) of sum(W[:,i:i]*H[i:i,:]) = colsum(W)_i * rowsum(H)_i  (0-based permutation)."""
    s = np.asarray(W, dtype=np.float32).sum(axis=0) * np.asarray(H, dtype=np.float32).sum(axis=1)
    return np.argsort(-s, kind="stable")


def getk(nkrange, robustness, cutoff=0.5, strict=True):
    """Post:7-41.  robustness: len(nkrange) values, or the full k-indexed vector (element k-1)."""
    nkrange = [int(k) for k in nkrange]
    r = np.asarray(robustness, dtype=np.float64)
    if len(r) != len(nkrange):
        r = r[[k - 1 for k in nkrange]]
    if np.all(np.isnan(r)):
        return 0
    if len(nkrange) == 1:
        if strict:
            return nkrange[-1] if r[-1] > cutoff else None
        return nkrange[-1]
    above = [i for i, v in enumerate(r) if v > cutoff]
    if not above:
        if strict:
            return None
        return nkrange[int(np.argmax(np.where(np.isnan(r), -np.inf, r)))]
    return nkrange[above[-1]]


def input_checks(X, load, save, casefilename, mixture, method, algorithm, clusterWmatrix, quiet=True):
    """Exec:95-175 for the part of the option space this path serves."""
    if load and casefilename == "":
        casefilename = "nmfk"
    if save and casefilename == "":
        casefilename = "nmfk"
    if mixture not in ("null", None):
        raise NotImplementedError("mixture != :null (MixMatch, Ipopt) is outside the :simple hot path")
    if getattr(X, "ndim", np.ndim(X)) > 2:
        raise ValueError("NMFk analysis can be executed for matrices!")  # ArgumentError, Exec:110-112
    method = str(method).lstrip(":")
    if method in _METHOD_ALIASES or method in ("nmf", "sparsity", "ipopt", "nlopt"):
        if not _is_sparse(X) and np.isnan(np.asarray(X, dtype=np.float32)).any() and method not in ("ipopt", "nlopt"):
            warnings.warn(f"Analyzed matrix has NaN's! NMF method {method} cannot be used! "
                          "Simple multiplicative NMF will be performed!")  # Exec:128-130
            method = "simple"
        else:
            raise NotImplementedError(f"method=:{method} is a different solver; libnmfk_hip implements method=:simple")
    if method != "simple":
        raise ValueError(f"Unknown method: {method}")  # Exec:777
    if X.ndim == 2 and X.shape[0] < X.shape[1] and not quiet:
        warnings.warn(f"Processed matrix size has more columns than rows (matrix size={X.shape})!")
    return load, save, casefilename, "null", method, algorithm, clusterWmatrix


def _mu_params(kw):
    """Peels the NMFmultiplicative keyword arguments (Mult:24, Exec:729) out of kw."""
    names = dict(tol="tol", tolOF="tolOF", maxiter="maxiter", maxreattempts="maxreattempts", maxbaditers="maxbaditers",
                 stopconv="stopconv", Wfixed="Wfixed", Hfixed="Hfixed", weight="weight")
    args = {}
    for key in list(kw):
        if key in names:
            args[names[key]] = kw.pop(key)
        elif key == "lambda_" or key == "lambda":
            args["lambda_"] = kw.pop(key)
    weight_array = None
    if "weight" in args and np.ndim(args["weight"]) != 0:
        weight_array = np.asarray(args.pop("weight"), dtype=np.float32)  # Mult:74 broadcast; shapes of Exec:484
    args["_weight_array"] = weight_array
    compute = kw.pop("compute", "f32")
    args["compute"] = {"f32": _lib.COMPUTE_F32, "f64": _lib.COMPUTE_F64}[compute]
    return args


def _sweep(ctx, X, ks, nNMF, kw, need_all_W=True):
    """All restarts of all ranks in `ks`: dict k -> dict(W (R,n,k), H (R,k,m), objvalue, iters, reason)."""
    from . import parallel

    kw = dict(kw)
    seed = kw.pop("seed", None)
    Winit, Hinit = kw.pop("Winit", None), kw.pop("Hinit", None)
    mu = _mu_params(kw)
    weight_array = mu.pop("_weight_array")
    normalizevector = kw.pop("normalizevector", None)
    if normalizevector is not None and len(normalizevector) == 0:
        normalizevector = None
    for junk in ("quiet", "veryquiet", "serial", "transpose", "scale", "bootstrap"):
        if junk in ("transpose", "scale", "bootstrap") and kw.get(junk):
            raise NotImplementedError(f"{junk}=true is outside the hot path (default off in the reference, Exec:729)")
        kw.pop(junk, None)
    if kw:
        # the reference swallows unknown keywords in NMFmultiplicative's kw... (Mult:24); be loud instead
        raise TypeError(f"unknown keyword arguments: {sorted(kw)}")
    modifymatrices = not (mu.get("Wfixed") or mu.get("Hfixed"))  # Exec:486-489 (haskey; we use truthiness)
    mu["normalize"] = int(modifymatrices)
    if seed is None:  # global-RNG path of the reference (Random.seed!(s) before execute): numpy's global RNG here;
        # drawn on rank 0 for everybody, so that a multi-rank sweep is one seed family
        seed = parallel.bcast_object(int(np.random.randint(0, 2 ** 31 - 1)))
    n, m = X.shape
    wi = hi = None
    if Winit is not None or Hinit is not None:
        if len(ks) != 1:
            raise ValueError("Winit/Hinit can only be given for a single rank")
        k = ks[0]
        if Winit is not None:
            Winit = np.asarray(Winit, dtype=np.float32)
            assert Winit.shape == (n, k), "size(Winit) == (n, k)"  # Mult:40
            if np.isnan(Winit).any():
                raise ValueError("Initial values for the W matrix entries include NaNs!")  # Mult:42-44
            wi = {k: np.broadcast_to(Winit, (nNMF, n, k))}
        if Hinit is not None:
            Hinit = np.asarray(Hinit, dtype=np.float32)
            assert Hinit.shape == (k, m), "size(Hinit) == (k, m)"  # Mult:50
            if np.isnan(Hinit).any():
                raise ValueError("Initial values for the H matrix entries include NaNs!")  # Mult:52-54
            hi = {k: np.broadcast_to(Hinit, (nNMF, k, m))}
    params = _lib.default_params(**mu)
    seeds = np.array([[run_seed(seed, k, r) for r in range(nNMF)] for k in ks], dtype=np.uint64)
    if normalizevector is not None and _is_sparse(X):
        raise NotImplementedError("normalizevector with a sparse X: scale the rows of X yourself")
    if weight_array is not None and _is_sparse(X):
        raise NotImplementedError("array-valued weight needs the dense path")
    if normalizevector is not None:  # Mult:27-31: X ./= normalizevector (rows) for the duration of the loop
        v = np.asarray(normalizevector, dtype=np.float32)
        if v.shape != (n,):
            raise ValueError(f"Length of normalizing vector does not match: {v.size} vs {n}")
        ctx.set_X((np.asarray(X, dtype=np.float32) / v[:, None]).astype(np.float32), mu.get("lambda_", 1e-32))
    ctx.set_weight(weight_array)
    comm = parallel.comm_of(ctx)
    if comm is not None:  # ranks joined through the C ABI: shard, RCCL all-gather of the device buffers
        lean = not (need_all_W or normalizevector is not None) and hasattr(comm, "bcast")
        res = comm.mu_sweep(ks, nNMF, seeds=seeds, Winit=wi, Hinit=hi, params=params, need_W=not lean)
        if lean:
            # best = true (Exec:655-658) needs ONE W per rank k besides the H of every restart: every rank knows every objective,
            # the owner of the best restart (Exec:545-546) broadcasts its W (nmfk_comm_bcast), the others stay local or absent
            for k in ks:
                best = int(np.argsort(res[k]["objvalue"], kind="stable")[0])
                owner, _slot = _lib.shard_owner(nNMF, comm.nranks, best)
                Wb = np.ascontiguousarray(res[k]["W"][best], dtype=np.float32)
                comm.bcast(Wb, owner)
                W = [np.asarray(res[k]["W"][r]) if r % comm.nranks == comm.rank else None for r in range(nNMF)]
                W[best] = Wb
                res[k]["W"] = W
    else:
        res = parallel.sharded_sweep(ctx.mu_sweep, ks, nNMF, seeds, wi, hi, params, n, m,
                                     need_all_W=need_all_W or normalizevector is not None)
    if normalizevector is not None:  # Mult:119-122: X .*= normalizevector; W .*= normalizevector, then Exec:791-792
        ctx.set_X(X, mu.get("lambda_", 1e-32))
        for k in ks:
            res[k]["W"] = np.ascontiguousarray(res[k]["W"] * v[None, :, None])
            res[k]["objvalue"] = np.array([ctx.frobenius(res[k]["W"][r], res[k]["H"][r]) for r in range(nNMF)],
                                          dtype=np.float32)
            # the library's sse belongs to the row-normalised X of the loop; the reference's "OF is very different" check
            # (Exec:602-607) works on the restored X with the rescaled W, i.e. on what objvalue now holds: nothing to compare
            res[k]["sse"] = None
    return res, params


def _nan_removed(post):
    """nanaction = :removed (Exec:581-595) drops a restart whose W OR H holds a NaN: every rank must see every W to reach the same
    selection, so the lean result exchange of a multi-GPU sweep (W stays with its owner) is not taken (a NaN in a foreign W
    would be seen by its owner only and idxsol, the clustering and the robustness would differ between the ranks)."""
    return str(post.get("nanaction", "zeroed")).lstrip(":") == "removed"


def _execute_run_post(ctx, X, nk, nNMF, res, clusterWmatrix=False, acceptratio=1, acceptfactor=math.inf, best=True,
                      nanaction="zeroed", quiet=True, saveall=False, resultdir=".", casefilename=""):
    """Everything of execute_run after the restart loop (Exec:545-710)."""
    n, m = X.shape
    # Matrix{T} (Exec:529-531).  Multi-GPU with best=true: only this rank's restarts and the best one carry a W
    WBig = [None if res["W"][i] is None else np.array(res["W"][i], dtype=np.float32) for i in range(nNMF)]
    HBig = [np.array(res["H"][i], dtype=np.float32) for i in range(nNMF)]
    objvalue = np.asarray(res["objvalue"], dtype=np.float32)
    idxsort = np.argsort(objvalue, kind="stable")  # Exec:545
    bestIdx = int(idxsort[0])
    Wbest, Hbest = WBig[bestIdx].copy(), HBig[bestIdx].copy()
    idxrat = np.ones(nNMF, dtype=bool)
    if acceptratio < 1:  # Exec:552-558
        ccc = int(math.ceil(nNMF * acceptratio))
        idxrat = np.array([True] * ccc + [False] * (nNMF - ccc))
        warnings.warn(f"NMF solutions removed based on an acceptance ratio: {idxrat.sum()} out of {nNMF} solutions remain")
    idxcut = np.ones(nNMF, dtype=bool)
    if acceptfactor < math.inf:  # Exec:559-565
        idxcut = objvalue[idxsort] < objvalue[bestIdx] * acceptfactor
        warnings.warn(f"NMF solutions removed based on an acceptance factor: {idxcut.sum()} out of {nNMF} solutions remain")
    idxnan = np.ones(nNMF, dtype=bool)
    nanaction = str(nanaction).lstrip(":")
    if nanaction == "zeroed":  # Exec:567-580
        zerod = 0
        for i in idxsort:
            isnh = np.isnan(HBig[i])
            HBig[i][isnh] = 0
            isnw = np.zeros(1, dtype=bool)
            if WBig[i] is not None:
                isnw = np.isnan(WBig[i])
                WBig[i][isnw] = 0
            zerod += bool(isnw.any() or isnh.any())
        if zerod:
            warnings.warn(f"NMF solutions contain NaN's: {zerod} out of {nNMF} solutions! NaN's have been converted to zeros!")
    elif nanaction == "removed":  # Exec:581-595
        for i in idxsort:
            if (WBig[i] is not None and np.isnan(WBig[i]).any()) or np.isnan(HBig[i]).any():
                idxnan[i] = False
    idxsol = idxrat & idxcut & idxnan  # Exec:596
    if idxsol.sum() < nNMF and not quiet:  # Exec:597-600
        print(f"NMF solutions removed based on various criteria: {idxsol.sum()} out of {nNMF} solutions remain")
    # Exec:602-607: of = normnan((X - W*H) .* weight) against the stored objective (which is unweighted, Exec:791): the
    # weighted residual norm of the final factors is the sqrt of the library's sse output (Mult:125)
    if res.get("sse") is not None:
        for i in range(nNMF):
            of = math.sqrt(max(float(res["sse"][i]), 0.0))
            if of > 0 and abs(of - float(objvalue[i])) / of > 1e-4:
                warnings.warn(f"OF {i + 1} is very different: {of} vs {objvalue[i]}!")
    sel = idxsort[idxsol]  # WBig[idxsort][idxsol]
    minsilhouette = 1.0
    extra = dict(objvalue=objvalue, idxsort=idxsort, iters=np.asarray(res["iters"]), reason=np.asarray(res["reason"]))
    Wa = Ha = None
    if nk > 1:
        Hs = np.stack([HBig[i] for i in sel])
        if clusterWmatrix:
            # Exec:621: clustersolutions(WBig[idxsort][idxsol], true) works on the W matrices THEMSELVES (Clus:426-428
            # makes no copies): the first solution's W is the running sum and ends up as the centroids (Clus:453-455,
            # 484, 512); everything downstream (Exec:633, finalize Fin:45-50) sees the mutated array.
            Wst = np.stack([WBig[i].T for i in sel])  # (nsol, k, n): signals as vectors of length n
            labels, centroids, _, _ = ctx.cluster_silhouette(Wst)
            WBig[sel[0]] = np.ascontiguousarray(centroids.T)
            Wst[0] = centroids
            psil, csil = ctx.silhouette(Wst, labels)
        else:
            labels, centroids, psil, csil = ctx.cluster_silhouette(Hs)  # Exec:623 + silhouettes of Exec:637
        Wb0, Hb0 = WBig[bestIdx], HBig[bestIdx]
        for i, c in enumerate(labels[:, 0]):  # Exec:631-635
            Wbest[:, i] = Wb0[:, c - 1]
            Hbest[i, :] = Hb0[c - 1, :]
        minsilhouette = float(np.min(csil))  # Exec:638
        extra.update(labels=labels, centroids=centroids, psil=psil, csil=csil)
        if not best or (saveall and casefilename != ""):
            Ws = np.stack([WBig[i] for i in sel])
            Wa, Ha, Wv, Hv = ctx.cluster_stats(Ws, Hs, labels)  # Fin:64-77
            extra.update(Wvar=Wv, Hvar=Hv)
    elif not best:
        # Exec:648 -> Fin:114-118: finalize(WBig[idxsol], HBig[idxsol]) masks the UNSORTED vectors with idxsol and takes
        # mean(Wa[1]; dims=2), mean(Ha[1]; dims=1) of the first survivor in restart order (not the lowest objective)
        first = int(np.flatnonzero(idxsol)[0])
        Wa = WBig[first].mean(axis=1, keepdims=True)
        Ha = HBig[first].mean(axis=0, keepdims=True)
    if saveall and casefilename != "":  # Exec:650-654: everything, before `best` replaces the cluster means
        if nk == 1:  # (the reference stops here with an UndefVarError: clustersilhouettes only exists for nk > 1)
            raise NameError("clustersilhouettes not defined (saveall needs nk > 1, src/NMFkExecute.jl:652)")
        fn = _result_filename(resultdir, casefilename, n, m, nk, nNMF, "-all")
        os.makedirs(resultdir, exist_ok=True)
        f32 = resultio.as_julia
        resultio.save(fn, **{"W": [f32(w) for w in WBig], "H": [f32(h) for h in HBig], "Wmean": f32(Wa), "Hmean": f32(Ha),
                             "Wvar": f32(extra["Wvar"]), "Hvar": f32(extra["Hvar"]), "Wbest": f32(Wbest), "Hbest": f32(Hbest),
                             "fit": f32(objvalue), "Cluster Silhouettes": f32(extra["csil"]),
                             "Cluster assignments": np.asarray(extra["labels"], dtype=np.int64),
                             "Cluster centroids": f32(extra["centroids"])})
        if not quiet:
            print(f"All results are saved in {fn}!")
    if best:
        Wa, Ha = Wbest, Hbest  # Exec:655-658
    phi_final = ctx.frobenius(Wa, Ha)  # Exec:664-667 (E[isnan] = 0 then norm == normnan)
    phi_final = float(np.float32(phi_final))
    numobservations = int(X.shape[0] * X.shape[1] - ctx.nan_count)  # Exec:697
    numparameters = Wa.size + Ha.size
    aic = 2 * numparameters + numobservations * math.log(phi_final / numobservations) if phi_final > 0 else -math.inf
    extra.update(Wbest=Wbest, Hbest=Hbest)
    return Wa, Ha, phi_final, minsilhouette, aic, extra


def _loadall(resultdir, casefilename, n, m, nk, nNMF):
    """Exec:499-509: the restarts of an earlier run from `<case>_<n>_<m>_<nk>_<nNMF>-all.jld` ("W", "H", "fit"), or None."""
    fn = _result_filename(resultdir, casefilename, n, m, nk, nNMF, "-all")
    if not os.path.isfile(fn):
        warnings.warn(f"File {fn} with ALL results is missing; runs will be executed!")
        return None
    z = resultio.load(fn, "W", "H", "fit")
    print(f"All results are loaded from {fn}!")
    R = len(z["W"])
    return dict(W=np.stack([np.asarray(w, np.float32) for w in z["W"]]), H=np.stack([np.asarray(h, np.float32) for h in z["H"]]),
                objvalue=np.asarray(z["fit"], np.float32), sse=None, iters=np.zeros(R, np.int32), reason=np.zeros(R, np.int32))


def execute_run(X, nk, nNMF, device=None, return_details=False, **kw):
    """execute_run(X, nk, nNMF; ...) (Exec:483-711) -> (Wa, Ha, phi_final, minsilhouette, aic)."""
    if not _is_sparse(X):
        X = np.asarray(X)
    if X.shape[0] * X.shape[1] == 0:
        raise ValueError(f"Input array has a zero dimension! Array size={X.shape}")
    ctx = _context(device)
    _upload(ctx, X, kw.get("lambda_", 1e-32))
    post = {k: kw.pop(k) for k in ("clusterWmatrix", "acceptratio", "acceptfactor", "best", "nanaction", "saveall", "resultdir",
                                   "casefilename") if k in kw}
    loadall = kw.pop("loadall", False)
    for k in ("mixture", "method", "algorithm"):
        kw.pop(k, None)
    n, m = X.shape
    res = None
    if loadall and post.get("casefilename", "") != "":  # Exec:499-509
        res = _loadall(post.get("resultdir", "."), post["casefilename"], n, m, int(nk), int(nNMF))
        if res is not None:
            post["saveall"] = False
    if res is None:
        need_all_W = bool(post.get("clusterWmatrix")) or not post.get("best", True) or bool(post.get("saveall")) or _nan_removed(post)
        res = _sweep(ctx, X, [int(nk)], int(nNMF), kw, need_all_W=need_all_W)[0][int(nk)]
    out = _execute_run_post(ctx, X, int(nk), int(nNMF), res, **post)
    return out if return_details else out[:5]


def hash_sha256_hex(X):
    """Exec:62-66.  The reference hashes Julia's Serialization stream of X; that framing is Julia-internal, so the
    digest here is over (dtype, shape, column-major little-endian bytes) -- the sidecar protocol is the same, the hex
    strings of the two implementations are not interchangeable."""
    h = hashlib.sha256()
    if _is_sparse(X):
        X = X.tocsc()
        h.update(f"csc {X.dtype.str} {X.shape}".encode())
        for a in (X.indptr, X.indices, X.data):
            h.update(np.ascontiguousarray(a).tobytes())
    else:
        X = np.asarray(X)
        h.update(f"{X.dtype.str} {X.shape}".encode())
        h.update(np.asfortranarray(X).tobytes(order="F"))
    return h.hexdigest()


def check_x_hash(X, xfile, quiet=True):
    """check_x_hash! (Exec:68-93): `<xfile>.sha256` sidecar; written when absent, compared (warning on mismatch)
    when present.  Returns the digest."""
    h = hash_sha256_hex(X)
    hashfile = xfile + ".sha256"
    if os.path.isfile(hashfile):
        with open(hashfile) as f:
            stored = f.read().strip()
        if stored and stored != h:
            warnings.warn(f"Matrix hash mismatch in '{hashfile}': Cached results may not correspond to this matrix! "
                          "Consider deleting the hash file and cached results to avoid confusion.")
        elif not quiet:
            print(f"Matrix hash DOES match the stored hash in '{hashfile}'.")
    else:
        os.makedirs(os.path.dirname(hashfile) or ".", exist_ok=True)
        with open(hashfile, "w") as f:
            f.write(h + "\n")
        if not quiet:
            print(f"Matrix hash saved in '{hashfile}'.")
    return h


def _result_filename(resultdir, casefilename, n, m, nk, nNMF, suffix=""):
    # Exec:265, 324: "<case>_<n>_<m>_<nk>_<nNMF>.jld" (Exec:500, 651: "...-all.jld" for the saveall payload)
    return os.path.join(resultdir, f"{casefilename}_{n}_{m}_{nk}_{nNMF}{suffix}{resultio.EXT}")


def execute(X, nkrange, nNMF=10, opts=None, *, cutoff=0.5, clusterWmatrix=False, mixture="null", method="simple",
            algorithm="multdiv", resultdir=".", load=True, save=True, casefilename="", loadonly=False, quiet=False,
            check_inputs=True, ordersignals=True, device=None, ctx=None, return_details=False, **kw):
    """NMFk.execute (Exec:178-233 for a range of k, Exec:236-329 for one k).

    nkrange: an int (-> 5-tuple W, H, fit, robustness, aic) or a range/list (-> 6-tuple with kopt).
    ctx: an nmfk Context on which set_X(X) has ALREADY been called (X resident in HBM, e.g. for repeated sweeps);
    by default the per-device context is used and X is uploaded here."""
    single = isinstance(nkrange, (int, np.integer))
    if opts is not None:  # the options-based overloads forward exactly these fields (Exec:33-47 range, Exec:50-65 one k)
        if not isinstance(opts, ExecuteOptions):
            raise TypeError("the fourth positional argument is an ExecuteOptions")
        o = opts
        fwd = dict(clusterWmatrix=o.clusterWmatrix, mixture=o.mixture, method=o.method, algorithm=o.algorithm,
                   resultdir=o.resultdir, load=o.load, save=o.save, casefilename=o.casefilename, dims=o.dims)
        fwd.update(dict(loadonly=o.loadonly, quiet=o.quiet, check_inputs=o.check_inputs, ordersignals=o.ordersignals) if single
                   else dict(cutoff=o.cutoff))
        return execute(X, nkrange, nNMF, device=device, ctx=ctx, return_details=return_details, **fwd, **kw)
    kw.pop("dims", None)  # Exec:178, 236: `dims` only matters for tensors (N > 2), which this path rejects
    if not _is_sparse(X):
        X = np.asarray(X)
    if X.ndim > 2:
        raise ValueError("NMFk analysis can be executed for matrices!")
    if X.shape[0] * X.shape[1] == 0:
        raise ValueError(f"Input array has a zero dimension! Array size={X.shape}")  # Exec:242-244
    ks = [int(nkrange)] if single else [int(k) for k in nkrange]
    if loadonly:  # Exec:245-251
        load, save = True, False
    load, save, casefilename, mixture, method, algorithm, clusterWmatrix = input_checks(
        X, load, save, casefilename, mixture, method, algorithm, clusterWmatrix, quiet=quiet)
    n, m = X.shape
    maxk = max(ks)
    W, H = [None] * maxk, [None] * maxk
    fitquality = np.zeros(maxk, dtype=np.float32)
    robustness = np.zeros(maxk, dtype=np.float32)
    aic = np.zeros(maxk, dtype=np.float32)
    fitquality[0], robustness[0] = np.inf, -1  # Exec:200-201
    details = {}
    post = {k: kw.pop(k) for k in ("acceptratio", "acceptfactor", "best", "nanaction", "saveall") if k in kw}
    loadall = kw.pop("loadall", False)
    if "Wfixed" in kw or "Hfixed" in kw:  # Exec:305-307
        ordersignals = False

    # Result cache (Exec:256-303).  With several ranks (one process per GPU) only rank 0 touches the files: it reads the
    # cached ranks and decides what is still to do, everybody else receives that decision -- every rank must enter the
    # sharded sweep with the SAME list of ranks -- and the loaded factors.
    from . import parallel

    rank0 = parallel.world()[0] == 0
    todo, recheck = [], []
    if rank0:
        if load or save:  # Exec:256-262 (the reference hashes X once per k; here once per call, when the cache is in use)
            xs = "_".join(str(v) for v in X.shape)
            xfile = os.path.join(resultdir, f"{casefilename or 'nmfk'}_x_matrix_{xs}{resultio.EXT}")
            if save and not single:  # Exec:185-192: the range form keeps the matrix next to its results, key "X"
                os.makedirs(resultdir, exist_ok=True)
                if _is_sparse(X):
                    # Deviation (stated in DESIGN.md section 8): the reference writes `X` itself, i.e. for a sparse input a Julia SparseMatrixCSC
                    # (a JLD compound type this writer does not produce).  The same information goes into the same file as the CSC's own fields
                    # under Julia's field names, 1-based like Julia stores them: SparseMatrixCSC(X_m, X_n, X_colptr, X_rowval, X_nzval)
                    Xc = X.tocsc()
                    Xc.sort_indices()
                    resultio.save(xfile, X_m=np.int64(Xc.shape[0]), X_n=np.int64(Xc.shape[1]), X_colptr=Xc.indptr.astype(np.int64) + 1,
                                  X_rowval=Xc.indices.astype(np.int64) + 1, X_nzval=np.asarray(Xc.data))
                else:
                    resultio.save(xfile, X=np.asfortranarray(X))
            check_x_hash(X, xfile, quiet=quiet)
        for nk in ks:  # Exec:264-303: per-k result cache
            fn = _result_filename(resultdir, casefilename, n, m, nk, nNMF)
            if load and not os.path.isfile(fn):  # Exec:266-269: old file-name convention
                old = os.path.join(resultdir, f"{casefilename}-{nk}-{nNMF}{resultio.EXT}")
                fn = old if os.path.isfile(old) else fn
            if load and os.path.isfile(fn):
                z = resultio.load(fn)
                Wl, Hl = np.asarray(z["W"]), np.asarray(z["H"])
                if Wl.shape == (n, nk) and Hl.shape == (nk, m):
                    W[nk - 1], H[nk - 1] = Wl, Hl
                    fitquality[nk - 1], robustness[nk - 1], aic[nk - 1] = z["fit"], z["robustness"], z["aic"]
                    recheck.append(nk)
                    continue
                if not quiet:  # Exec:287-289
                    print(f"File {fn} contains inconsistent results; runs will be executed ...")
            if loadonly:  # Exec:291-298 sentinel
                W[nk - 1], H[nk - 1] = np.zeros((0, 0), np.float32), np.zeros((0, 0), np.float32)
                fitquality[nk - 1], robustness[nk - 1], aic[nk - 1] = np.inf, -1, -np.inf
                continue
            todo.append(nk)
    if parallel.world()[1] > 1:
        todo, recheck, W, H, fitquality, robustness, aic = parallel.bcast_object((todo, recheck, W, H, fitquality, robustness, aic))

    if ctx is None and (todo or not all(np.isinf(fitquality[[k - 1 for k in ks]]))):
        ctx = _context(device)
        _upload(ctx, X, kw.get("lambda_", 1e-32))  # raises "All matrix entries must be nonnegative!" (Mult:4-7)
    for nk in recheck:  # Exec:274-283: a loaded result whose fit does not match X is re-saved with the new fit
        fit = ctx.frobenius(W[nk - 1], H[nk - 1])
        if abs(fit - fitquality[nk - 1]) > np.finfo(np.float16).eps:
            warnings.warn(f"Fit quality is not consistent: {fit} != {fitquality[nk - 1]}")
            fitquality[nk - 1] = fit
            if rank0:
                resultio.save(_result_filename(resultdir, casefilename, n, m, nk, nNMF), W=W[nk - 1], H=H[nk - 1],
                              fit=fitquality[nk - 1], robustness=robustness[nk - 1], aic=aic[nk - 1])
    if todo:
        res = {}
        if loadall and casefilename != "":  # Exec:499-509 per rank: restarts of an earlier run instead of new ones
            for nk in todo:
                r = _loadall(resultdir, casefilename, n, m, nk, int(nNMF)) if rank0 else None
                r = parallel.bcast_object(r)
                if r is not None:
                    res[nk] = r
        run = [nk for nk in todo if nk not in res]
        if run:
            need_all_W = bool(clusterWmatrix) or not post.get("best", True) or bool(post.get("saveall")) or _nan_removed(post)
            res.update(_sweep(ctx, X, run, int(nNMF), kw, need_all_W=need_all_W)[0])
        for nk in todo:
            sv = dict(post, saveall=bool(post.get("saveall")) and rank0 and nk in run)
            Wa, Ha, phi, sil, a, extra = _execute_run_post(ctx, X, nk, int(nNMF), res[nk], clusterWmatrix, quiet=quiet,
                                                           resultdir=resultdir, casefilename=casefilename, **sv)
            so = signalorder(Wa, Ha) if ordersignals else np.arange(nk)  # Exec:311-318
            W[nk - 1], H[nk - 1] = Wa[:, so], Ha[so, :]
            fitquality[nk - 1], robustness[nk - 1], aic[nk - 1] = phi, sil, a
            extra["signalorder"] = so
            details[nk] = extra
            if not quiet:  # Exec:322
                print("Signals: %2d Fit: %12.7g Silhouette: %12.7g AIC: %12.7g Signal order: %s" % (nk, phi, sil, a, so + 1))
            if save and rank0:  # Exec:323-327 (written to a temporary name and renamed: readers never see a partial file)
                os.makedirs(resultdir, exist_ok=True)
                resultio.save(_result_filename(resultdir, casefilename, n, m, nk, nNMF), W=W[nk - 1], H=H[nk - 1],
                              fit=fitquality[nk - 1], robustness=robustness[nk - 1], aic=aic[nk - 1])
        if save:
            parallel.barrier()  # nobody starts the next call before rank 0 has finished writing
    if single:
        nk = ks[0]
        out = (W[nk - 1], H[nk - 1], fitquality[nk - 1], robustness[nk - 1], aic[nk - 1])
        return out + (details.get(nk),) if return_details else out

    if np.all(np.isinf(fitquality[[k - 1 for k in ks]])):  # Exec:206-208
        warnings.warn("No successful NMFk runs!")
        kopt = 0
    else:
        for nk in ks:  # Exec:211-224
            fit = ctx.frobenius(W[nk - 1], H[nk - 1]) if W[nk - 1].size else np.inf
            if abs(fit - fitquality[nk - 1]) > np.finfo(np.float16).eps:
                warnings.warn(f"Fit quality is not consistent: {fit} != {fitquality[nk - 1]}")
            fitquality[nk - 1] = fit
            if not quiet:
                print("Signals: %2d Fit: %12.7g Silhouette: %12.7g AIC: %12.7g" % (nk, fitquality[nk - 1],
                                                                                   robustness[nk - 1], aic[nk - 1]))
        kopt = getk(ks, robustness[[k - 1 for k in ks]], cutoff)  # Exec:225
    out = (W, H, fitquality, robustness, aic, kopt)
    return out + (details,) if return_details else out
