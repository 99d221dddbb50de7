"""Sharding of the flat (k, restart) work list over the GPUs of one node, one process per GPU.

The product path is the C ABI: `attach(ctx)` joins the rank's Context to an RCCL communicator inside libnmfk_hip
(nmfk_comm_create) and `Comm.mu_sweep` = nmfk_mu_sweep_sharded does the shard / all-gather on device buffers.
The torch.distributed functions below (`broadcast_X`, `sharded_sweep`) are the same plan in Python: the CPU rehearsal of
the N > 1 logic (gloo, tests/test_parallel_gloo.py) and the fallback when no communicator is attached.

The reference's only parallelism is `Distributed.pmap` over the restarts of ONE k (src/NMFkExecute.jl:511-526),
shipping X to a worker with every task.  Here the (k, restart) units of the WHOLE sweep are sharded (`plan_shards`: rank g owns restarts {g, g+N, ...} of
every k), X is broadcast once, nothing is exchanged inside the MU loop, and the results (H stack, objective,
iterations, and W or only the best W per k) are exchanged in ONE padded all-gather each at the end, so that every
rank can run the clustering step."""
import numpy as np

_comms = {}  # id(Context) -> _lib.Comm: ranks joined through the C ABI (nmfk_comm_*, RCCL inside libnmfk_hip)


def attach(ctx):
    """Joins this process' Context to the job's RCCL communicator THROUGH THE C ABI (nmfk_comm_create; the data path of
    bench.py --gpus N and of execute() under torchrun).  torch.distributed (already initialised, any backend) only
    carries the 128-byte unique id from rank 0 to the others."""
    from . import _lib

    d = _dist()
    if d is None:
        return None
    if id(ctx) in _comms:
        return _comms[id(ctx)]
    box = [_lib.comm_unique_id() if d.get_rank() == 0 else None]
    d.broadcast_object_list(box, src=0)
    comm = _lib.Comm(ctx, d.get_world_size(), d.get_rank(), box[0])
    _comms[id(ctx)] = comm
    return comm


class _MultiAdapter:
    """`Comm.mu_sweep`'s interface on a _lib.Multi (one process, one host thread per rank inside libnmfk_hip)."""

    def __init__(self, multi):
        self.multi = multi

    def mu_sweep(self, ks, nruns, seeds=None, Winit=None, Hinit=None, params=None, need_W=True, **kw):
        return self.multi.mu_sweep(ks, nruns, seeds=seeds, Winit=Winit, Hinit=Hinit, params=params, **kw)

    def close(self):
        pass


def attach_multi(multi):
    """execute(..., ctx=multi.ctx0) then runs its sweeps through nmfk_multi_sweep: the restarts sharded over the ranks of the
    Multi (GPUs, or the logical ranks of the loopback transport), results delivered to the caller, clustering on rank 0's
    context.  The one-process form of the N > 1 path (`NMFkHIP.execute(...; ngpus = N)` on the Julia side)."""
    _comms[id(multi.ctx0)] = _MultiAdapter(multi)
    return multi.ctx0


def comm_of(ctx):
    return _comms.get(id(ctx))


def detach(ctx):
    c = _comms.pop(id(ctx), None)
    if c is not None:
        c.close()


def bcast_object(obj, src=0):
    """Host-side object from rank `src` to everybody (cache decisions, seeds); identity without a process group."""
    d = _dist()
    if d is None:
        return obj
    box = [obj if d.get_rank() == src else None]
    d.broadcast_object_list(box, src=src)
    return box[0]


def barrier():
    d = _dist()
    if d is not None:
        d.barrier()


def _dist():
    import sys

    if "torch" not in sys.modules:  # nobody in this process can have initialised a process group (and importing torch
        return None                 # here would cost seconds and load a second copy of librccl next to libnmfk_hip's)
    try:
        import torch.distributed as dist
    except Exception:  # torch without distributed support: single process
        return None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def broadcast_X(X, src=0):
    """RCCL broadcast of X from rank `src` (the pmap closure capture of Exec:516 done once)."""
    d = _dist()
    if d is None:
        return np.asarray(X, dtype=np.float32)
    import torch

    dev = _device(d)
    shape = torch.tensor(list(np.shape(X)) if d.get_rank() == src else [0, 0], dtype=torch.int64, device=dev)
    d.broadcast(shape, src)
    n, m = (int(v) for v in shape.tolist())
    if d.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(dev)
    else:
        t = torch.empty((n, m), dtype=torch.float32, device=dev)
    d.broadcast(t, src)
    return t.cpu().numpy()


def _device(d):
    import torch

    if d.get_backend() == "nccl":
        import os

        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cpu")


def _all_gather_np(d, a):
    """all_gather of equally-shaped numpy arrays -> list over ranks."""
    import torch

    dev = _device(d)
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    outs = [torch.empty_like(t) for _ in range(d.get_world_size())]
    d.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def plan_shards(ks, nruns, world):
    """Deterministic assignment of the (k, restart) units to ranks (identical on every rank): rank g owns the restarts
    {g, g + world, ...} of EVERY k.  The cost of a unit grows with k and with its iteration count, which is only
    known afterwards, so giving every rank the same mix of ranks balances the load where dealing out whole ranks
    does not (measured on MI355X at 8 ranks: cost-balanced blocks of 16 restarts of ~4 ranks per GPU and 4 restarts
    of all 15 ranks per GPU take the same time, see scripts/rank_sim.py).
    Returns (c, chunks): every rank runs c = ceil(nruns / world) restarts of every k (short lists are padded by
    repeating the last restart; the padding runs are dropped); chunks = [(kidx, restarts, owner)]."""
    c = -(-nruns // world)
    chunks = []
    for q in range(len(ks)):
        for g in range(world):
            rs = list(range(g, nruns, world))
            if rs:
                chunks.append((q, rs, g))
    return c, chunks


def _exchange(d, payload, sizes):
    """All-gather of one byte string per rank (lengths `sizes`, known everywhere) -> list of uint8 arrays."""
    import torch

    dev = _device(d)
    cap = max(max(sizes), 1)
    buf = np.zeros(cap, dtype=np.uint8)
    buf[:payload.size] = payload
    t = torch.from_numpy(buf).to(dev)
    outs = [torch.empty_like(t) for _ in range(d.get_world_size())]
    d.all_gather(outs, t)
    return [o.cpu().numpy()[:sizes[g]] for g, o in enumerate(outs)]


_FIELDS = (("H", np.float32), ("objvalue", np.float32), ("sse", np.float64), ("iters", np.int32), ("reason", np.int32))


def sharded_sweep(sweep_fn, ks, nruns, seeds, Winit, Hinit, params, n, m, need_all_W=True):
    """Runs `sweep_fn` (Context.mu_sweep) on this rank's chunks and returns the results of ALL restarts.

    seeds: (len(ks), nruns).  Result: dict k -> dict(W (nruns,n,k), H (nruns,k,m), objvalue, sse, iters, reason).
    need_all_W=False (the default `best=true`, clusterWmatrix=false path of execute_run, Exec:655-658): only the W of
    the restart with the lowest objective is exchanged; W of the other restarts is then only present for this rank's
    own restarts, None elsewhere ("W" becomes a list)."""
    d = _dist()
    if d is None:
        return sweep_fn(ks, nruns, seeds=seeds, Winit=Winit, Hinit=Hinit, params=params)
    rank, N = d.get_rank(), d.get_world_size()
    ks = [int(k) for k in ks]
    c, chunks = plan_shards(ks, nruns, N)
    seeds = np.asarray(seeds)
    mine = [ch for ch in chunks if ch[2] == rank]
    local = {}
    if mine:
        pad = {q: rs + [rs[-1]] * (c - len(rs)) for q, rs, _ in mine}
        lks = [ks[q] for q, *_ in mine]
        sub = lambda dct: None if dct is None else {ks[q]: np.asarray(dct[ks[q]])[pad[q]] for q in pad if dct.get(ks[q]) is not None}
        local = sweep_fn(lks, c, seeds=np.stack([seeds[q, pad[q]] for q, *_ in mine]), Winit=sub(Winit), Hinit=sub(Hinit),
                         params=params)
    fields = _FIELDS + ((("W", np.float32),) if need_all_W else ())

    def nbytes(ch):
        k, cnt = ks[ch[0]], len(ch[1])
        return cnt * ((k * m) * 4 + 4 + 8 + 4 + 4 + ((n * k) * 4 if need_all_W else 0))

    parts = []
    for q, rs, _ in mine:
        for key, dt in fields:
            parts.append(np.ascontiguousarray(np.asarray(local[ks[q]][key])[:len(rs)], dtype=dt).reshape(-1).view(np.uint8))
    sizes = [sum(nbytes(ch) for ch in chunks if ch[2] == g) for g in range(N)]
    got = _exchange(d, np.concatenate(parts) if parts else np.zeros(0, np.uint8), sizes)
    out = {k: dict(H=np.empty((nruns, k, m), np.float32), objvalue=np.empty(nruns, np.float32), sse=np.empty(nruns, np.float64),
                   iters=np.empty(nruns, np.int32), reason=np.empty(nruns, np.int32)) for k in ks}
    if need_all_W:
        for k in ks:
            out[k]["W"] = np.empty((nruns, n, k), np.float32)
    owner, slot = {}, {}
    offs = [0] * N
    for q, rs, g in chunks:  # same order as the packing loops of every rank
        k, cnt = ks[q], len(rs)
        for key, dt in fields:
            shape = {"H": (cnt, k, m), "W": (cnt, n, k)}.get(key, (cnt,))
            nb = int(np.prod(shape)) * np.dtype(dt).itemsize
            out[k][key][rs] = got[g][offs[g]:offs[g] + nb].view(dt).reshape(shape)
            offs[g] += nb
        for j, r in enumerate(rs):
            owner[(q, r)], slot[(q, r)] = g, j
    if not need_all_W:
        best = {q: int(np.argsort(out[ks[q]]["objvalue"], kind="stable")[0]) for q in range(len(ks))}  # Exec:545-546
        wsz = [sum(n * ks[q] * 4 for q in best if owner[(q, best[q])] == g) for g in range(N)]
        wb = [np.ascontiguousarray(local[ks[q]]["W"][slot[(q, best[q])]], dtype=np.float32).reshape(-1).view(np.uint8)
              for q in sorted(best) if owner[(q, best[q])] == rank]
        gotw = _exchange(d, np.concatenate(wb) if wb else np.zeros(0, np.uint8), wsz)
        offs = [0] * N
        for q in sorted(best):
            k, g = ks[q], owner[(q, best[q])]
            W = [None] * nruns
            for qq, rs, _ in mine:
                if qq == q:
                    for j, r in enumerate(rs):
                        W[r] = np.asarray(local[k]["W"][j])
            nb = n * k * 4
            W[best[q]] = gotw[g][offs[g]:offs[g] + nb].view(np.float32).reshape(n, k).copy()
            offs[g] += nb
            out[k]["W"] = W
    return out
