"""ctypes binding of libnmfk_hip.so (include/nmfk_hip.h).  There is NO CPU fallback: every entry point needs
the HIP library and a gfx950 GPU, and fails loudly otherwise."""
import ctypes as C
import os
import subprocess

import numpy as np

# The sweep runs one stream per rank k; ROCm maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4),
# and kernels that share a queue serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ.setdefault("NMFK_STREAMS", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMFK_HIP_LIB", os.path.join(_HERE, "libnmfk_hip.so"))  # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nmfk_hip.h")

NMFK_OK = 0
ERR_BAD_ARG, ERR_NEGATIVE, ERR_NAN_INIT, ERR_NO_X, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_RCCL = 1, 2, 3, 4, 5, 6, 7, 8
UNIQUE_ID_BYTES = 128
STOP_MAXITER, STOP_STAGNATION, STOP_TOL, STOP_CONSISTENCY = 1, 2, 3, 4
COMPUTE_F32, COMPUTE_F64 = 0, 1
MAX_K = 64


class NMFkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class MuParams(C.Structure):
    """nmfk_mu_params (include/nmfk_hip.h): keyword arguments of NMFmultiplicative, src/NMFkMultiplicative.jl:24."""
    _fields_ = [("tol", C.c_double), ("tolOF", C.c_double), ("lambda_", C.c_double), ("weight", C.c_double),
                ("maxiter", C.c_int64), ("maxreattempts", C.c_int32), ("maxbaditers", C.c_int32),
                ("stopconv", C.c_int32), ("Wfixed", C.c_int32), ("Hfixed", C.c_int32), ("normalize", C.c_int32),
                ("compute", C.c_int32), ("reserved", C.c_int32)]


def build(force=False, verbose=False):
    """Compile libnmfk_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libnmfk_hip.so failed")
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library (symbols only: no GPU is touched until a context is created)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NMFkError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, fp, dp, ip, i64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)
    L.nmfk_version.restype = C.c_int
    L.nmfk_last_error.restype = C.c_char_p
    L.nmfk_device_count.argtypes = [C.POINTER(C.c_int)]
    L.nmfk_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nmfk_destroy.argtypes = [vp]
    L.nmfk_device_info.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), i64p]
    L.nmfk_mu_default_params.argtypes = [C.POINTER(MuParams)]
    L.nmfk_set_X.argtypes = [vp, fp, C.c_int64, C.c_int64, C.c_int64, C.c_double, i64p, i64p]
    L.nmfk_fill_uniform.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64, fp]
    pp = C.POINTER(C.c_void_p)
    L.nmfk_mu_sweep.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_int, pp, pp, C.POINTER(C.c_uint64),
                                C.POINTER(MuParams), pp, pp, pp, pp, pp, pp]
    L.nmfk_mu_batch.argtypes = [vp, C.c_int, C.c_int, fp, fp, C.POINTER(C.c_uint64), C.POINTER(MuParams), fp, fp, fp,
                                dp, ip, ip]
    L.nmfk_cluster_silhouette.argtypes = [vp, C.c_int, C.c_int, C.c_int64, fp, ip, fp, fp, fp]
    L.nmfk_set_X_csc.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, fp, i64p]
    L.nmfk_silhouette.argtypes = [vp, C.c_int, C.c_int, C.c_int64, fp, ip, fp, fp]
    L.nmfk_set_weight.argtypes = [vp, fp, C.c_int64, C.c_int64]
    L.nmfk_cluster_stats.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int64, fp, fp, ip, fp, fp, fp, fp]
    L.nmfk_robustkmeans_ex.argtypes = [vp, C.c_int, C.c_int64, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, ip, fp, fp,
                                       ip, C.POINTER(C.c_double), ip, ip, ip, C.c_void_p, C.c_void_p, ip]
    L.nmfk_robustkmeans.argtypes = L.nmfk_robustkmeans_ex.argtypes[:-1]  # (ABI 200's signature, kept: no `converged`)
    L.nmfk_frobenius.argtypes = [vp, C.c_int, fp, fp, C.POINTER(C.c_double)]
    L.nmfk_set_profiling.argtypes = [vp, C.c_int]
    L.nmfk_set_objective_trace.argtypes = [vp, C.c_int]
    L.nmfk_get_objective_trace.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.nmfk_last_sweep_info.argtypes = [vp, C.POINTER(C.c_int32)]
    L.nmfk_last_sweep_info_ex.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
    i32p = C.POINTER(C.c_int32)
    L.nmfk_shard_plan.argtypes = [C.c_int, C.c_int, C.c_int, i32p, i32p]
    L.nmfk_comm_bcast.argtypes = [vp, C.c_int, C.c_void_p, C.c_int64]
    L.nmfk_shard_owner.argtypes = [C.c_int, C.c_int, C.c_int, i32p, i32p]
    L.nmfk_plan_hyb_tiers.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.POINTER(C.c_int)]
    L.nmfk_loopback_group_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nmfk_loopback_group_destroy.argtypes = [vp]
    L.nmfk_comm_create_loopback.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.nmfk_multi_create_loopback.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.nmfk_multi_comm.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.nmfk_comm_unique_id.argtypes = [C.c_void_p]
    L.nmfk_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_void_p, C.POINTER(vp)]
    L.nmfk_comm_destroy.argtypes = [vp]
    L.nmfk_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.nmfk_comm_bcast_X.argtypes = [vp, C.c_int, fp, C.c_int64, C.c_int64, C.c_int64, C.c_double, i64p, i64p, i64p, i64p]
    L.nmfk_mu_sweep_sharded.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int32), C.c_int, pp, pp, C.POINTER(C.c_uint64),
                                        C.POINTER(MuParams), C.c_int, pp, pp, pp, pp, pp, pp]
    L.nmfk_multi_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nmfk_multi_destroy.argtypes = [vp]
    L.nmfk_multi_context.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.nmfk_multi_set_X.argtypes = [vp, fp, C.c_int64, C.c_int64, C.c_int64, C.c_double, i64p, i64p]
    L.nmfk_multi_sweep.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_int, pp, pp, C.POINTER(C.c_uint64),
                                   C.POINTER(MuParams), pp, pp, pp, pp, pp, pp]
    L.nmfk_get_profile.argtypes = [vp, C.c_int, C.c_void_p, C.POINTER(C.c_double), i64p, C.POINTER(C.c_double),
                                   C.POINTER(C.c_int)]
    _lib = L
    return L


def _check(rc):
    if rc != NMFK_OK:
        raise NMFkError(rc, lib().nmfk_last_error().decode("utf-8", "replace"))


def default_params(**kw):
    p = MuParams()
    _check(lib().nmfk_mu_default_params(C.byref(p)))
    for key, val in kw.items():
        name = "lambda_" if key in ("lambda", "lambda_") else key
        if not hasattr(p, name):
            raise TypeError(f"unknown MU parameter {key!r}")
        cur = getattr(p, name)
        setattr(p, name, type(cur)(val) if not isinstance(val, bool) else int(val))
    return p


def device_count():
    c = C.c_int(0)
    _check(lib().nmfk_device_count(C.byref(c)))
    return c.value


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _sweep_call(call, n, m, ks, nruns, seeds, Winit, Hinit, params, kw, need_W=None, collective=False):
    """Marshals the arguments of nmfk_mu_sweep / nmfk_mu_sweep_sharded / nmfk_multi_sweep and unpacks the results."""
    if not (n and m):  # (the result buffers are sized from them: a wrapper that was never given X must not reach the library)
        if not collective:
            raise NMFkError(ERR_NO_X, "nmfk_set_X has not been called through this object (set_X / set_X_sparse)")
        # a per-rank collective call: leaving here would strand the other ranks in the status agreement of
        # nmfk_mu_sweep_sharded.  Enter the library with 1 x 1 result buffers instead: its local plan fails with
        # NMFK_ERR_NO_X before anything is written, and agree() hands that status to every rank.
        n = m = 1
    P = params if params is not None else default_params(**kw)
    ks = [int(k) for k in ks]
    nk = len(ks)
    arr_k = (C.c_int32 * nk)(*ks)
    PP = C.c_void_p * nk
    keep = []

    def table(d, shape_of):
        t = PP()
        any_ = False
        for q, k in enumerate(ks):
            a = None if d is None else d.get(k)
            if a is None:
                t[q] = None
                continue
            a = np.asarray(a, dtype=np.float32)
            if a.shape != shape_of(k):
                raise NMFkError(ERR_BAD_ARG, f"initial factor for k={k} has shape {a.shape}, expected {shape_of(k)}")
            # natural (r, row, col) -> stacked column-major: (r, col, row) C-contiguous
            a = np.ascontiguousarray(np.transpose(a, (0, 2, 1)))
            keep.append(a)
            t[q] = a.ctypes.data
            any_ = True
        return t if any_ else None

    wi = table(Winit, lambda k: (nruns, n, k))
    hi = table(Hinit, lambda k: (nruns, k, m))
    sd = None
    if seeds is not None:
        sd_np = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(nk, nruns))
        keep.append(sd_np)
        sd = sd_np.ctypes.data_as(C.POINTER(C.c_uint64))
    out = {}
    tw, th, tf, ts, ti, tr = PP(), PP(), PP(), PP(), PP(), PP()
    for q, k in enumerate(ks):
        o = dict(Wt=np.full((nruns, k, n), np.nan, dtype=np.float32), Ht=np.empty((nruns, m, k), dtype=np.float32),
                 objvalue=np.empty(nruns, dtype=np.float32), sse=np.empty(nruns, dtype=np.float64),
                 iters=np.empty(nruns, dtype=np.int32), reason=np.empty(nruns, dtype=np.int32))
        out[k] = o
        tw[q], th[q], tf[q] = o["Wt"].ctypes.data, o["Ht"].ctypes.data, o["objvalue"].ctypes.data
        ts[q], ti[q], tr[q] = o["sse"].ctypes.data, o["iters"].ctypes.data, o["reason"].ctypes.data
    args = [nk, arr_k, int(nruns), wi, hi, sd, C.byref(P)]
    if need_W is not None:
        args.append(int(need_W))
    _check(call(*args, tw, th, tf, ts, ti, tr))
    for k, o in out.items():
        o["W"] = np.transpose(o.pop("Wt"), (0, 2, 1))  # views: (nruns, n, k), Fortran-ordered per restart
        o["H"] = np.transpose(o.pop("Ht"), (0, 2, 1))  # (nruns, k, m)
    return out


def shard_plan(nruns, nranks, rank):
    """nmfk_shard_plan -> (real restarts of the shard, padded count every shard runs)."""
    cnt, pad = C.c_int32(), C.c_int32()
    _check(lib().nmfk_shard_plan(int(nruns), int(nranks), int(rank), C.byref(cnt), C.byref(pad)))
    return cnt.value, pad.value


def shard_owner(nruns, nranks, r):
    """nmfk_shard_owner -> (rank that runs restart r, its local slot): r = rank + slot * nranks."""
    g, j = C.c_int32(), C.c_int32()
    _check(lib().nmfk_shard_owner(int(nruns), int(nranks), int(r), C.byref(g), C.byref(j)))
    return g.value, j.value


def plan_hyb_tiers(n, m, variant, units, cus=256):
    """nmfk_plan_hyb_tiers (host arithmetic, no device) -> one dict per tier of the retire-aware schedule: units and, for the
    H and the W half-step, {res, wsplit, S, dchunk, fused, slots, ns}; cohorts.  variant 4 / 8 / 16: units of that kernel variant only;
    0: the ranks 2..16 in equal numbers (the bench sweep's mix)."""
    cap = 40
    out = (C.c_int32 * (16 * cap))()
    cnt = C.c_int()
    _check(lib().nmfk_plan_hyb_tiers(int(n), int(m), int(variant), int(units), int(cus), out, cap, C.byref(cnt)))
    names = ("res", "wsplit", "S", "dchunk", "fused", "slots", "ns")
    rows = []
    for j in range(cnt.value):
        o = out[16 * j:16 * j + 16]
        rows.append({"units": o[0], "H": dict(zip(names, o[1:8])), "W": dict(zip(names, o[8:15])), "cohorts": o[15]})
    return rows


def _pin_rccl():
    """One copy of RCCL per process: inside a PyTorch process libnmfk_hip must use the librccl PyTorch ships and has
    loaded (two copies double-free at exit), so point the library's dlopen at it before its first RCCL call."""
    import sys

    if "NMFK_RCCL_LIB" in os.environ or "torch" not in sys.modules:
        return
    cand = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "librccl.so")
    if os.path.exists(cand):
        os.environ["NMFK_RCCL_LIB"] = cand


def comm_unique_id():
    """nmfk_comm_unique_id: 128 opaque bytes, generated on ONE rank and handed to the others by the host layer."""
    _pin_rccl()
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(lib().nmfk_comm_unique_id(buf))
    return bytes(buf.raw)


class Comm:
    """nmfk_comm: RCCL communicator of one rank (one Context = one GPU); collective calls, one process per GPU."""

    def __init__(self, ctx, nranks, rank, unique_id):
        self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
        self._h = C.c_void_p()
        _pin_rccl()
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise NMFkError(ERR_BAD_ARG, f"the RCCL unique id has {UNIQUE_ID_BYTES} bytes")
        idbuf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        _check(lib().nmfk_comm_create(ctx._h, self.nranks, self.rank, idbuf, C.byref(self._h)))
        ctx._comms = getattr(ctx, "_comms", []) + [self]  # the context closes its communicators before itself

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value and getattr(self, "_owned", True):
            lib().nmfk_comm_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close

    def bcast_X(self, X=None, root=0, lambda_=1e-32):
        """nmfk_comm_bcast_X: X is read on `root` only; every rank ends with X resident (NMFpreprocessing! done)."""
        n = m = 0
        ptr = None
        if self.rank == root:
            Xf = np.asfortranarray(X, dtype=np.float32)
            n, m = Xf.shape
            ptr = Xf.ctypes.data
        no, mo, nan, zero = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _check(lib().nmfk_comm_bcast_X(self._h, int(root), ptr, n, m, max(n, 1), float(lambda_), C.byref(no), C.byref(mo),
                                       C.byref(nan), C.byref(zero)))
        self.ctx.n, self.ctx.m = no.value, mo.value
        self.ctx.nan_count, self.ctx.zero_count = nan.value, zero.value
        return self.ctx

    def bcast(self, arr, root):
        """nmfk_comm_bcast: the C-contiguous array `arr` of rank `root` arrives in `arr` of every other rank (in place)."""
        assert arr.flags["C_CONTIGUOUS"] and arr.flags["WRITEABLE"]
        _check(lib().nmfk_comm_bcast(self._h, int(root), arr.ctypes.data, int(arr.nbytes)))
        return arr

    def mu_sweep(self, ks, nruns, seeds=None, Winit=None, Hinit=None, params=None, need_W=True, **kw):
        """nmfk_mu_sweep_sharded: same arguments on every rank (ALL restarts); every rank gets all results.  need_W=False:
        W comes back for this rank's own restarts only (NaN elsewhere)."""
        return _sweep_call(lambda *a: lib().nmfk_mu_sweep_sharded(self.ctx._h, self._h, *a), self.ctx.n, self.ctx.m, ks, nruns,
                           seeds, Winit, Hinit, params, kw, need_W=bool(need_W), collective=True)


class Multi:
    """nmfk_multi: one process, GPUs 0..ngpus-1 (a context, a communicator and a host thread per GPU).
    loopback=True (test hook): `ngpus` LOGICAL ranks on the one GPU `device`, collectives emulated by device copies --
    the N > 1 code of the C ABI on a one-GPU box (nmfk_multi_create_loopback)."""

    def __init__(self, ngpus, loopback=False, device=0):
        self._h = C.c_void_p()
        if loopback:
            _check(lib().nmfk_multi_create_loopback(int(ngpus), int(device), C.byref(self._h)))
        else:
            _pin_rccl()
            _check(lib().nmfk_multi_create(int(ngpus), C.byref(self._h)))
        self.ngpus = int(ngpus)
        self.ctx0 = self.context(0)  # GPU 0's context (not owned): clustering, silhouettes, fit re-checks

    def context(self, g):
        h = C.c_void_p()
        _check(lib().nmfk_multi_context(self._h, int(g), C.byref(h)))
        ctx = Context.__new__(Context)
        ctx._h, ctx._owned = h, False
        ctx.n = ctx.m = ctx.nan_count = ctx.zero_count = 0
        return ctx

    def comm(self, g):
        """rank g's communicator as a Comm (not owned): per-rank collective calls from the caller's own threads."""
        h = C.c_void_p()
        _check(lib().nmfk_multi_comm(self._h, int(g), C.byref(h)))
        c = Comm.__new__(Comm)
        c.ctx, c.nranks, c.rank = self.context(g), self.ngpus, int(g)
        c.ctx.n, c.ctx.m = self.ctx0.n, self.ctx0.m
        c._h, c._owned = h, False
        return c

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().nmfk_multi_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_X(self, X, lambda_=1e-32):
        Xf = np.asfortranarray(X, dtype=np.float32)
        n, m = Xf.shape
        nan, zero = C.c_int64(), C.c_int64()
        _check(lib().nmfk_multi_set_X(self._h, Xf.ctypes.data, n, m, max(n, 1), float(lambda_), C.byref(nan), C.byref(zero)))
        self.ctx0.n, self.ctx0.m, self.ctx0.nan_count, self.ctx0.zero_count = n, m, nan.value, zero.value
        return self

    def set_X_sparse(self, X):
        """Sparse X on every rank (nmfk_set_X_csc takes host pointers, so every rank's context is given the matrix; there
        is no device-side broadcast of the CSC arrays)."""
        for g in range(self.ngpus):
            c = self.context(g).set_X_sparse(X)
        self.ctx0.n, self.ctx0.m, self.ctx0.nan_count, self.ctx0.zero_count = c.n, c.m, 0, c.zero_count
        self.ctx0.nnz = c.nnz
        return self

    def mu_sweep(self, ks, nruns, seeds=None, Winit=None, Hinit=None, params=None, **kw):
        return _sweep_call(lambda *a: lib().nmfk_multi_sweep(self._h, *a), self.ctx0.n, self.ctx0.m, ks, nruns, seeds, Winit,
                           Hinit, params, kw)


class Context:
    """One GPU (nmfk_ctx).  Arrays cross the boundary as float32 numpy arrays in Julia (column-major) layout:
    a stack of R matrices n x k is passed as a C-contiguous array of shape (R, k, n)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().nmfk_create(int(device), C.byref(self._h)))
        self.n = self.m = 0
        self.nan_count = self.zero_count = 0

    def close(self):
        for c in getattr(self, "_comms", []):
            c.close()
        self._comms = []
        if getattr(self, "_h", None) is not None and self._h.value and getattr(self, "_owned", True):
            lib().nmfk_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(), C.c_int64()
        _check(lib().nmfk_device_info(self._h, name, 256, C.byref(cus), C.byref(mem)))
        return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=mem.value)

    def set_X(self, X, lambda_=1e-32):
        """NMFpreprocessing! (Mult:3-22).  X: (n, m) array (any layout); NaN = missing.  The caller's array is
        not modified."""
        X = np.asarray(X)
        if X.ndim != 2:
            raise NMFkError(ERR_BAD_ARG, "NMFk analysis can be executed for matrices!")  # Exec:110-112
        Xf = np.asfortranarray(X, dtype=np.float32)
        n, m = Xf.shape
        nan, zero = C.c_int64(), C.c_int64()
        _check(lib().nmfk_set_X(self._h, Xf.ctypes.data, n, m, max(n, 1), float(lambda_), C.byref(nan), C.byref(zero)))
        self.n, self.m = n, m
        self.nan_count, self.zero_count = nan.value, zero.value
        return self

    def set_X_sparse(self, X):
        """nmfk_set_X_csc: X is a scipy.sparse matrix (any format); zeros stay zeros (gather kernels)."""
        Xc = X.tocsc()
        Xc.sum_duplicates()
        n, m = Xc.shape
        colptr = np.ascontiguousarray(Xc.indptr, dtype=np.int64)
        rowidx = np.ascontiguousarray(Xc.indices, dtype=np.int32)
        vals = np.ascontiguousarray(Xc.data, dtype=np.float32)
        kept = C.c_int64()
        _check(lib().nmfk_set_X_csc(self._h, n, m, len(vals), colptr.ctypes.data, rowidx.ctypes.data, vals.ctypes.data,
                                    C.byref(kept)))
        self.n, self.m = n, m
        self.nan_count, self.zero_count = 0, n * m - kept.value
        self.nnz = kept.value
        return self

    def set_X_csc_raw(self, n, m, colptr, rowidx, vals):
        """nmfk_set_X_csc with the caller's arrays AS THEY ARE (no canonical form): what a direct user of the C ABI passes."""
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowidx = np.ascontiguousarray(rowidx, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float32)
        kept = C.c_int64()
        _check(lib().nmfk_set_X_csc(self._h, int(n), int(m), len(vals), colptr.ctypes.data, rowidx.ctypes.data, vals.ctypes.data,
                                    C.byref(kept)))
        self.n, self.m = int(n), int(m)
        self.nan_count, self.zero_count = 0, self.n * self.m - kept.value
        self.nnz = kept.value
        return self

    def fill_uniform(self, seed, offset, count):
        out = np.empty(count, dtype=np.float32)
        _check(lib().nmfk_fill_uniform(self._h, C.c_uint64(seed), C.c_uint64(offset), count, out.ctypes.data))
        return out

    def mu_sweep(self, ks, nruns, seeds=None, Winit=None, Hinit=None, params=None, **kw):
        """nmfk_mu_sweep.  ks: list of ranks; seeds: (len(ks), nruns) uint64; Winit/Hinit: optional dicts
        k -> array (nruns, n, k) / (nruns, k, m) in natural (row, col) indexing.
        Returns dict k -> dict(W (nruns, n, k), H (nruns, k, m), objvalue (nruns,) float32, sse, iters, reason)."""
        return _sweep_call(lambda *a: lib().nmfk_mu_sweep(self._h, *a), self.n, self.m, ks, nruns, seeds, Winit, Hinit,
                           params, kw)

    def cluster_silhouette(self, Hs):
        """nmfk_cluster_silhouette.  Hs: (nsol, k, m) sorted by objective.  Returns labels (k, nsol) int32 1-based,
        centroids (k, m), point silhouettes (k, nsol), cluster silhouettes (k,)."""
        Hs = np.asarray(Hs, dtype=np.float32)
        nsol, k, m = Hs.shape
        stack = np.ascontiguousarray(np.transpose(Hs, (0, 2, 1)))  # (nsol, m, k): column-major k x m per solution
        labels = np.empty((nsol, k), dtype=np.int32)
        cent = np.empty((m, k), dtype=np.float32)
        psil = np.empty((nsol, k), dtype=np.float32)
        csil = np.empty(k, dtype=np.float32)
        _check(lib().nmfk_cluster_silhouette(self._h, k, nsol, m, stack.ctypes.data, labels.ctypes.data,
                                             cent.ctypes.data, psil.ctypes.data, csil.ctypes.data))
        return labels.T, cent.T, psil.T, csil

    def silhouette(self, Hs, labels):
        """nmfk_silhouette: silhouettes of the stack Hs (nsol, k, len) for GIVEN labels (k, nsol)."""
        Hs = np.asarray(Hs, dtype=np.float32)
        nsol, k, m = Hs.shape
        stack = np.ascontiguousarray(np.transpose(Hs, (0, 2, 1)))
        lab = np.ascontiguousarray(np.asarray(labels, dtype=np.int32).T)
        psil = np.empty((nsol, k), dtype=np.float32)
        csil = np.empty(k, dtype=np.float32)
        _check(lib().nmfk_silhouette(self._h, k, nsol, m, stack.ctypes.data, lab.ctypes.data, psil.ctypes.data,
                                     csil.ctypes.data))
        return psil.T, csil

    def set_weight(self, weight):
        """Array-valued `weight` of the monitored objective (Mult:74): scalar handled by MuParams.weight; a vector of
        length n or an (n, m) / (1, m) array is broadcast like Julia's `.*` against the residual.  None clears."""
        if weight is None:
            _check(lib().nmfk_set_weight(self._h, None, 0, 0))
            return
        w = np.asarray(weight, dtype=np.float32)
        if w.ndim == 1:
            if w.shape[0] != self.n:
                raise NMFkError(ERR_BAD_ARG, "length(weight) == size(X, 1)")  # Exec:484
            w = w[:, None]
        w = np.asfortranarray(np.broadcast_to(w, (self.n, self.m)))
        _check(lib().nmfk_set_weight(self._h, w.ctypes.data, self.n, self.m))

    def cluster_stats(self, Ws, Hs, labels):
        """nmfk_cluster_stats (Fin:64-77).  Ws (nsol, n, k), Hs (nsol, k, m), labels (k, nsol)."""
        Ws, Hs = np.asarray(Ws, dtype=np.float32), np.asarray(Hs, dtype=np.float32)
        nsol, n, k = Ws.shape
        m = Hs.shape[2]
        wst = np.ascontiguousarray(np.transpose(Ws, (0, 2, 1)))
        hst = np.ascontiguousarray(np.transpose(Hs, (0, 2, 1)))
        lab = np.ascontiguousarray(np.asarray(labels, dtype=np.int32).T)
        Wm, Wv = np.empty((k, n), np.float32), np.empty((k, n), np.float32)
        Hm, Hv = np.empty((m, k), np.float32), np.empty((m, k), np.float32)
        _check(lib().nmfk_cluster_stats(self._h, k, nsol, n, m, wst.ctypes.data, hst.ctypes.data, lab.ctypes.data,
                                        Wm.ctypes.data, Hm.ctypes.data, Wv.ctypes.data, Hv.ctypes.data))
        return Wm.T, Hm.T, Wv.T, Hv.T

    def robustkmeans(self, X, k, repeats=1000, maxiter=1000, tol=1e-32, seed=0, compute_silhouettes_flag=False):
        """nmfk_robustkmeans (Clus:172-246).  X: d x n, columns = samples.  Returns a dict (assignments 1-based, sorted
        by decreasing cluster size) and, with compute_silhouettes_flag, the point silhouettes of the best run."""
        Xf = np.asfortranarray(X, dtype=np.float32)
        d, n = Xf.shape
        assign, counts = np.empty(n, np.int32), np.empty(k, np.int32)
        centers, costs = np.empty((d, k), np.float32, order="F"), np.empty(n, np.float32)
        allc = np.empty(repeats, np.float64)
        sil = np.empty(n, np.float32) if compute_silhouettes_flag else None
        tc = C.c_double()
        br, it, kf, cv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        I = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        F = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _check(lib().nmfk_robustkmeans_ex(self._h, d, n, F(Xf), int(k), int(repeats), int(maxiter), float(tol), C.c_uint64(seed),
                                          I(assign), F(centers), F(costs), I(counts), C.byref(tc), C.byref(br), C.byref(it),
                                          C.byref(kf), allc.ctypes.data, None if sil is None else sil.ctypes.data, C.byref(cv)))
        # centers / counts: the clusters found; centers_k / counts_k: all k columns / entries as Clustering.KmeansResult holds
        # them (zero columns / counts for clusters that were not found)
        res = dict(assignments=assign, centers=centers[:, :kf.value], costs=costs, counts=counts[:kf.value], totalcost=tc.value,
                   iterations=it.value, converged=bool(cv.value), best_repeat=br.value, all_costs=allc, nclusters=kf.value,
                   centers_k=centers, counts_k=counts)
        return (res, sil) if compute_silhouettes_flag else res

    def frobenius(self, W, H):
        """normnan(X - W*H) (Help:226-228)."""
        W = np.asfortranarray(W, dtype=np.float32)
        H = np.asfortranarray(H, dtype=np.float32)
        if W.shape[0] != self.n or H.shape[1] != self.m or W.shape[1] != H.shape[0]:
            return float("inf")  # Exec:213-215: size mismatch => fit = Inf
        out = C.c_double()
        _check(lib().nmfk_frobenius(self._h, W.shape[1], W.ctypes.data, H.ctypes.data, C.byref(out)))
        return out.value

    def last_sweep_info(self):
        """nmfk_last_sweep_info: the launch schedule the last mu_sweep on this context took."""
        info = (C.c_int32 * 16)()
        _check(lib().nmfk_last_sweep_info_ex(self._h, info, 16))
        return dict(phases=info[0], mfma_group_units=info[1], merged_valu_groups=info[2], launch_groups=info[3],
                    wide_mfma_units=info[4], replans=info[5], last_tier=info[6], units_in_last_plan=info[7],
                    deferred_checks=info[8], plain_checks=info[9], cohorts=info[10], fused_reductions=info[11])

    def set_objective_trace(self, on=True):
        """nmfk_set_objective_trace: record the monitored objective (Mult:74) at every check of the next sweeps."""
        _check(lib().nmfk_set_objective_trace(self._h, int(on)))

    def objective_trace(self, kidx, restart, cap=100000):
        """nmfk_get_objective_trace: the checks of restart `restart` of the kidx-th rank of the last mu_sweep."""
        out = (C.c_double * cap)()
        n = C.c_int()
        _check(lib().nmfk_get_objective_trace(self._h, int(kidx), int(restart), out, cap, C.byref(n)))
        return np.array(out[:min(n.value, cap)], dtype=np.float64)

    def set_profiling(self, on=True):
        _check(lib().nmfk_set_profiling(self._h, int(on)))

    def get_profile(self):
        nmax = 80
        names = ((C.c_char * 64) * nmax)()
        ms = (C.c_double * nmax)()
        launches = (C.c_int64 * nmax)()
        flops = (C.c_double * nmax)()
        cnt = C.c_int()
        _check(lib().nmfk_get_profile(self._h, nmax, C.cast(names, C.c_void_p), ms, launches, flops, C.byref(cnt)))
        return {names[i].value.decode(): dict(ms=ms[i], launches=launches[i], flops=flops[i]) for i in range(cnt.value)}
