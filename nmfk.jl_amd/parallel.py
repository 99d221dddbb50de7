"""Sharding of the flat (k, restart) work list over the GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference's only parallelism is `Distributed.pmap` over the restarts of ONE k (src/NMFkExecute.jl:511-526),
shipping X to a worker with every task.  Here rank g owns restarts {g, g+N, ...} of EVERY k (cost per unit is
proportional to k x iterations, so sharding by restart balances the ranks where sharding by k would not), X is
broadcast once, nothing is exchanged inside the MU loop, and the per-k results (H stack, objective, and W) are
all-gathered once at the end so that every rank can run the clustering step."""
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # torch absent: single process
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


def sharded_sweep(sweep_fn, ks, nruns, seeds, Winit, Hinit, params, n, m, need_all_W=True):
    """Runs `sweep_fn` (Context.mu_sweep) on this rank's restarts and returns the results of ALL restarts.

    seeds: (len(ks), nruns).  Result: dict k -> dict(W (nruns,n,k), H (nruns,k,m), objvalue, sse, iters, reason).
    need_all_W=False (the default `best=true`, clusterWmatrix=false path of execute_run, Exec:655-658): only the W of
    the restart with the lowest objective is exchanged (one broadcast of n x k per rank k from its owner); W of the
    other restarts is then only present for this rank's own restarts, None elsewhere ("W" becomes a list)."""
    d = _dist()
    if d is None:
        return sweep_fn(ks, nruns, seeds=seeds, Winit=Winit, Hinit=Hinit, params=params)
    import torch

    rank, N = d.get_rank(), d.get_world_size()
    mine = list(range(rank, nruns, N))
    per = (nruns + N - 1) // N  # every rank runs `per` restarts so that all gathers have equal shapes;
    pad = mine + [mine[-1] if mine else 0] * (per - len(mine))  # padding restarts repeat one and are dropped
    sub = lambda dct: None if dct is None else {k: np.asarray(v)[pad] for k, v in dct.items()}
    local = sweep_fn(ks, per, seeds=np.asarray(seeds)[:, pad], Winit=sub(Winit), Hinit=sub(Hinit), params=params)
    dev = _device(d)
    out = {}
    for k in ks:
        o = {}
        keys = ("W", "H", "objvalue", "sse", "iters", "reason") if need_all_W else ("H", "objvalue", "sse", "iters", "reason")
        for key in keys:
            parts = _all_gather_np(d, np.ascontiguousarray(local[k][key]))
            full = np.empty((nruns,) + parts[0].shape[1:], dtype=parts[0].dtype)
            for g in range(N):
                idx = list(range(g, nruns, N))
                full[idx] = parts[g][:len(idx)]
            o[key] = full
        if not need_all_W:
            best = int(np.argsort(o["objvalue"], kind="stable")[0])  # Exec:545-546 (NaN sorts last)
            owner = best % N
            wb = np.ascontiguousarray(local[k]["W"][best // N]) if rank == owner else np.empty((n, k), np.float32)
            t = torch.from_numpy(wb).to(dev)
            d.broadcast(t, owner)
            W = [None] * nruns
            for j, r in enumerate(mine):
                W[r] = local[k]["W"][j]
            W[best] = t.cpu().numpy()
            o["W"] = W
        out[k] = o
    return out
