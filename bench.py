#!/usr/bin/env python3
"""bench.py -- NMF factorizations/sec on the BASELINE.json configuration, one rank per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete robustness sweep = NMFk.execute(X, 2:16, 32; method=:simple): 480 multiplicative-update
factorizations with the reference's default stop rule, clustering + silhouettes per k, and kopt.  X (dense U(0,1)
fp32, 8192 x 512) is resident in HBM before the timed region.  With N ranks the 480 factorizations are sharded by
restart (strong scaling: total work fixed), X is broadcast once over RCCL and the per-k results are all-gathered.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # one HW queue per concurrently running rank group (before HIP init)

import numpy as np

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBPS = 8000.0


TRAFFIC_FILES = ("profiles/r06/traffic.json", "profiles/r06/traffic_h_step.json", "profiles/r05/traffic.json", "profiles/r05/traffic_h_step.json", "profiles/r04/traffic.json", "profiles/r03/traffic.json", "profiles/r02/traffic.json")


def _pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel: FETCH_SIZE x 2 + WRITE_SIZE from separate `rocprofv3 --pmc` passes of
    this command (scripts/profile_bench.sh, scripts/make_traffic.py).  PMC passes cannot run inside the timed bench, so
    this is a STATIC figure read from the newest committed file -- the line says which (`traffic_source`) -- and only a file
    whose `kernel` IS the kernel this run timed counts (`kernel`: a name like "hyb_res_kernel"; a file of another kernel, or
    without the field, is refused: the figure would go stale silently when the kernels change).
    -> (bytes or None, source string)"""
    refused = []
    for rel in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                t = json.load(fh)
        except Exception:
            continue
        if not kernel or kernel not in str(t.get("kernel", "")):
            refused.append(f"{rel} (kernel {t.get('kernel', '?')!r})")
            continue
        return float(t["hbm_bytes_per_launch"]), (f"{rel} (static: rocprofv3 --pmc passes of an earlier run of this command, kernel "
                                                  f"{t['kernel']})")
    return None, "absent for kernel " + repr(kernel) + ("; refused: " + ", ".join(refused) if refused else "")


def julia_reference_baseline(n, m, ks, nruns):
    """the reference itself, if the box has `julia` and an importable NMFk.  Bounded: a
    fixed-budget sweep (maxbaditers=10^9, maxiter=20) of 2 restarts at k = min, max.  Returns a dict for the bench line
    ("julia": "absent" on this image -- SURVEY.md probe table)."""
    import shutil
    import subprocess

    exe = shutil.which("julia")
    if exe is None:
        return {"julia": "absent"}
    prog = (f"import Random; import NMFk; X = rand(Float32, {n}, {m}); "
            f"NMFk.execute(rand(Float32, 15, 5), 2:3, 2; method=:simple, load=false, save=false, quiet=true); "  # JIT warm-up
            f"t = @elapsed for k in ({ks[0]}, {ks[-1]}); NMFk.execute(X, k, 2; method=:simple, load=false, save=false, quiet=true, "
            f"maxiter=20, maxbaditers=10^9); end; println(\"NMFK_REF_SECONDS=\", t)")
    try:
        r = subprocess.run([exe, "-e", prog], capture_output=True, text=True, timeout=600, cwd=os.environ.get("TMPDIR", "/tmp"))
    except Exception as e:  # noqa: BLE001
        return {"julia": exe, "error": repr(e)}
    for ln in r.stdout.splitlines():
        if ln.startswith("NMFK_REF_SECONDS="):
            sec = float(ln.split("=")[1])
            return {"julia": exe, "kind": "reference", "seconds_for_4_factorizations_of_20_iterations": sec,
                    "sec_per_iteration_mean_of_kmin_kmax": sec / 80.0}
    return {"julia": exe, "error": "NMFk.jl is not importable on this box: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}


def cpu_baseline(X, ks, nruns, iters_by_k, threads):
    """Times the CPU oracle (C + OpenMP port of the reference's Float64 loop; oracle/nmfk_oracle.c) on this host
    for a bounded sample -- a fixed budget of MU iterations at k = min, mid, max -- and extrapolates to the whole
    sweep with the per-restart iteration counts of the GPU run: the reference publishes no timing
    and Julia is not installed, so this is a 'port' baseline, stated as such."""
    import nmfk_oracle as oracle

    oracle.build()
    n, m = X.shape
    sample_ks = sorted({ks[0], ks[len(ks) // 2], ks[-1]})
    budget = 20
    per_iter = {}
    for k in sample_ks:
        W0, H0 = oracle.init_factors(1, n, m, k)
        oracle.multiplicative(X, k, W0, H0, maxiter=2, maxbaditers=10 ** 9, nthreads=threads)  # warm the thread team
        t = time.perf_counter()
        oracle.multiplicative(X, k, W0, H0, maxiter=budget, maxbaditers=10 ** 9, nthreads=threads)
        per_iter[k] = (time.perf_counter() - t) / budget
    kk = np.array(sample_ks, dtype=np.float64)
    tt = np.array([per_iter[k] for k in sample_ks])
    slope, icpt = np.polyfit(kk, tt, 1) if len(kk) > 1 else (0.0, tt[0])
    total = sum(float(np.sum(iters_by_k[k])) * max(icpt + slope * k, 1e-9) for k in ks)
    return dict(value=len(ks) * nruns / total, unit="factorizations/s", cores=threads, kind="port",
                sample=f"{budget} MU iterations each at k={sample_ks} on the same X (fp64, as the reference's loop), "
                       f"{threads} OpenMP threads; per-iteration cost fitted linearly in k and multiplied by the GPU "
                       f"sweep's own per-restart iteration counts",
                sec_per_iter={str(k): per_iter[k] for k in sample_ks}, extrapolated_sweep_seconds=total)


def planted_matrix(ctx, n, m, k0=6, seed=2, noise=0.01, scale=1.0):
    """X = scale * (W0 H0 + noise * U), W0 in U(0,1)^{n x k0}, H0 in U(0,1)^{k0 x m}."""
    W0 = ctx.fill_uniform(seed, 0, n * k0).reshape(k0, n).T.astype(np.float64)
    H0 = ctx.fill_uniform(seed, n * k0, k0 * m).reshape(m, k0).T.astype(np.float64)
    U = ctx.fill_uniform(seed, n * k0 + k0 * m, n * m).reshape(m, n).T.astype(np.float64)
    return np.asfortranarray((scale * (W0 @ H0 + noise * U)).astype(np.float32))


def secondary_cfg4(NMFk, ctx, iters=100):
    """BASELINE configs[3]: sparse 0.5 %-fill fp32 X 100000 x 4096, k = 2:32, nruns = 16 (496 factorizations), a fixed budget
    of `iters` MU iterations with the check block every 10th.  Algorithmic bytes: 2*nnz*8 B + 4*(n+m)*k*4 B per
    iteration and factorization; time = GPU time of the MU loop (HIP events)."""
    import scipy.sparse as sp

    n, m, fill, R = 100000, 4096, 0.005, 16
    rng = np.random.default_rng(3)
    nnz = int(n * m * fill)
    Xs = sp.csc_matrix((rng.uniform(1, 5, nnz).astype(np.float32), (rng.integers(0, n, nnz), rng.integers(0, m, nnz))), shape=(n, m))
    Xs.sum_duplicates()
    ctx.set_profiling(False)
    ctx.set_X_sparse(Xs)
    ks = list(range(2, 33))
    seeds = np.array([[NMFk.run_seed(1, k, r) for r in range(R)] for k in ks], dtype=np.uint64)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=10, maxbaditers=10 ** 9)
    ctx.set_profiling(True)
    ctx.mu_sweep(ks, R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
    loop = ctx.get_profile()["mu_loop"]
    ctx.set_profiling(False)
    ms = loop["ms"] / iters
    bytes_iter = float(sum((2 * ctx.nnz * 8 + 4 * (n + m) * k * 4) * R for k in ks))
    gbps = bytes_iter / (ms * 1e-3) / 1e9
    return {"workload": f"sparse {fill:.1%}-fill fp32 X {n}x{m} ({ctx.nnz} non-zeros), k=2:32, nruns={R}: {len(ks) * R} factorizations, "
                        f"{iters} MU iterations (fixed budget, objective + check block every 10th)",
            "ms_per_iter": ms, "GBps_algorithmic": gbps, "frac_hbm": gbps / PEAK_HBM_GBPS, "bound": "hbm", "peak_GBps": PEAK_HBM_GBPS,
            "algorithmic_bytes_per_iter": bytes_iter,
            # aggregate HBM traffic (static: PMC passes cannot run inside the bench): 4.50 GB per launch x 8.44 launches per iteration, launches overlapping
            # 2.5-fold on their streams -- the chip's rate is bytes per iteration / ms per iteration, NOT the per-launch figure
            "hbm_bytes_per_iter_measured": 4.50e9 * 8.44, "hbm_GBps_measured": 4.50e9 * 8.44 / (ms * 1e-3) / 1e9,
            "hbm_measured_source": "profiles/r06/traffic_sp_blk.json x profiles/r06/secondary_kernel_stats.csv (844 launches per 100 iterations); round 6 changed the "
                                   "kernel's LDS side only (bank conflicts: profiles/r06/sparse_analysis.txt), not its global traffic",
            "kernel": "sp_blk_kernel<NC> (nmfk_step_impl.h; sliced ELL, one lane element per lane, the gathered factor through LDS) "
                      "for both half-steps of every rank <= 32",
            "profile": "profiles/r05/secondary_kernel_stats.csv (rocprofv3 --kernel-trace --stats of scripts/secondary.py)"}


def secondary_cfg2(NMFk, ctx, X, iters=1000):
    """BASELINE configs[1]: the bench matrix, k = 8, ONE restart (the single-GPU MU kernel on its own).  One factorization is
    8 workgroups of the resident W half-step and 128 of the H half-step: it cannot fill 256 CUs, the figure is a latency."""
    n, m, k = X.shape[0], X.shape[1], 8
    ctx.set_profiling(False)
    ctx.set_X(X)
    seeds = np.array([[NMFk.run_seed(1, k, 0)]], dtype=np.uint64)
    ctx.mu_sweep([k], 1, seeds=seeds, maxiter=20, maxbaditers=10 ** 9)
    ctx.set_profiling(True)
    ctx.mu_sweep([k], 1, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
    loop = ctx.get_profile()["mu_loop"]
    ctx.set_profiling(False)
    ms = loop["ms"] / iters
    return {"workload": f"dense U(0,1) fp32 X {n}x{m}, k={k}, nruns=1, {iters} MU iterations (fixed budget, check block every 10th)",
            "ms_per_iter": ms, "TFLOPs_algorithmic": 8.0 * n * m * k / (ms * 1e-3) / 1e12,
            "seconds_per_factorization_of_10000_iterations": ms * 10.0,
            "note": "a single factorization occupies a few per cent of the GPU (latency-bound); the metric's regime is the batched sweep"}


def secondary_cfg5(NMFk, ctx, iters=40):
    """BASELINE configs[4] shape: dense fp32 X 65536 x 2048, k = 64, 8 restarts (one GPU's share of the 64), fixed budget.
    Algorithmic flops 8*n*m*k per iteration and factorization over the GPU time of the MU loop (objective + check block
    every 10th iteration included)."""
    n, m, k, R = 65536, 2048, 64, 8
    X = ctx.fill_uniform(4, 0, n * m).reshape(m, n).T
    ctx.set_profiling(False)
    ctx.set_X(X)
    del X
    seeds = np.array([[NMFk.run_seed(1, k, r) for r in range(R)]], dtype=np.uint64)
    ctx.mu_sweep([k], R, seeds=seeds, maxiter=2, maxbaditers=10 ** 9)
    ctx.set_profiling(True)
    ctx.mu_sweep([k], R, seeds=seeds, maxiter=iters, maxbaditers=10 ** 9)
    prof = ctx.get_profile()
    ctx.set_profiling(False)
    loop = prof["mu_loop"]
    ms = loop["ms"] / iters
    tf = 8.0 * n * m * k * R / (ms * 1e-3) / 1e12
    halves = {nm: {"avg_launch_ms": v["ms"] / v["launches"], "TFLOPs": v["flops"] / (v["ms"] * 1e-3) / 1e12}
              for nm, v in prof.items() if nm.startswith(("h_step", "w_step")) and v["launches"]}
    # Roofs.  Round 6: at 48 / 64 signals BOTH products run on the bf16 matrix pipe from exact three-term splits (six term pairs each: 12 + 12
    # v_mfma_f32_16x16x32_bf16 of 16 cycles per 16 x 16 tile and 16 loop steps = 384 matrix cycles; rounds 3-5: 12 bf16 + 16 fp32 of 32 = 704; the
    # all-fp32 formulation the fp32 peak is quoted for: 1024).  The roof of THIS formulation is the dense bf16 peak / 6 term pairs; the fraction of it
    # is the matrix pipe's busy fraction at the peak clock (cross-checked by SQ_VALU_MFMA_BUSY_CYCLES, profiles/r06/wide_pmc_summary_bn1_final.txt).  The
    # algorithmic rate may exceed the fp32-MFMA peak -- that ratio is reported as a ratio, not as a fraction of a roof.
    bn = os.environ.get("NMFK_WIDE_BN", "1") != "0"
    cyc_per_tile = 384.0 if bn else 704.0
    roof = PEAK_FP32_TFLOPS * 1024.0 / cyc_per_tile  # fp32 peak x (1024 all-fp32 matrix cycles / cycles this form issues) = bf16 peak / 6 when bn
    return {"workload": f"dense U(0,1) fp32 X {n}x{m}, k={k}, {R} restarts, {iters} MU iterations (fixed budget, objective + check block every 10th)",
            "ms_per_iter": ms, "TFLOPs_algorithmic": tf, "bound": "mfma", "peak_TFLOPs": roof,
            "peak_note": ("dense bf16 MFMA peak (2516.8 = 16 x the fp32 peak) / 6 term pairs of the exact three-term splits both products run in"
                          if bn else "fp32 peak x 1024 / 704: 12 bf16 + 16 fp32 matrix instructions per tile instead of 32 fp32"),
            "frac": tf / roof, "matrix_pipe_busy": tf / roof,
            "matrix_pipe_busy_note": f"= algorithmic rate x {cyc_per_tile:.0f} matrix cycles per tile and chunk / (algorithmic flops per tile and chunk x SIMDs x 2.4 GHz); "
                                     "counters: profiles/r06/wide_pmc_summary_bn1_final.txt (SQ_VALU_MFMA_BUSY_CYCLES 1.61e9 per launch of 1 024 SIMDs, 3.03e6 cycles per launch)",
            "x_fp32_mfma_peak": tf / PEAK_FP32_TFLOPS,
            "x_fp32_mfma_peak_note": "ratio to the roof of an all-fp32-MFMA formulation (157.3): above 1 because the products run as bf16 splits at 3/8 of its matrix cycles",
            "half_steps": halves,
            "kernel": ("wide2_step_kernel<4,2,0,true> (nmfk_step_hyb.hip: W*H AND the numerators from exact three-term bf16 splits on v_mfma_f32_16x16x32_bf16, "
                       "explicit software pipeline; <4,2,2,true> behind a check iteration: the same half-step leaves the monitored objective)") if bn else
                      "wide2_step_kernel<4,2,0,false> (W*H from three-term bf16 splits, numerators in fp32 MFMAs)",
            "profile": "profiles/r06/secondary_kernel_stats.csv (rocprofv3 --kernel-trace --stats of scripts/secondary.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--kmin", type=int, default=2)
    ap.add_argument("--kmax", type=int, default=16)
    ap.add_argument("--nruns", type=int, default=32)
    ap.add_argument("--maxiter", type=int, default=10000)
    ap.add_argument("--compute", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-kopt-check", action="store_true", help="skip the planted-matrix 'same kopt' sweeps after the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg4 (sparse) / cfg5 (k = 64) measurements after the timed region")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("NMFK_BENCH_BUDGET_S", "400")),
                    help="wall seconds the whole run may take: the measurements BEHIND the timed region (same-kopt sweeps, secondary "
                         "workloads) run in order of importance while the run stays inside it, the rest is reported as skipped")
    ap.add_argument("--loopback", action="store_true",
                    help="REHEARSAL of the N > 1 path on ONE GPU: --gpus N logical ranks in this one process through libnmfk_hip's "
                         "loopback transport (nmfk_multi_create_loopback): shard plan, padding, status agreement, all-gather layout "
                         "and delivery are the code RCCL runs; the number is not a multi-GPU measurement")
    args = ap.parse_args()
    t_start = time.perf_counter()

    import torch
    import nmfk_jl_amd as NMFk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnmfk_hip has no CPU fallback")
    backend = os.environ.get("NMFK_DIST_BACKEND", "nccl")  # "gloo" only to rehearse the N>1 path on a 1-GPU box
    if os.environ.get("NMFK_FORCE_DEVICE"):
        local = int(os.environ["NMFK_FORCE_DEVICE"])
        os.environ["LOCAL_RANK"] = str(local)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or (args.loopback and world == 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"

    ks = list(range(args.kmin, args.kmax + 1))
    multi = None
    if args.loopback and args.gpus > 1:
        multi = NMFk.Multi(args.gpus, loopback=True, device=local)
        ctx = NMFk.parallel.attach_multi(multi)
    else:
        ctx = NMFk.Context(local)
    # synthetic X: U(0,1) from the library's portable generator (identical on every rank: the host copy only gives
    # execute() its shape); the DEVICE copy every rank computes on comes from rank 0 over RCCL
    X = np.asfortranarray(ctx.fill_uniform(1, 0, args.n * args.m).reshape(args.m, args.n).T)
    comm = None
    bcast_prof = None
    if world > 1 and backend == "nccl":
        # the N > 1 data path is the C ABI's: nmfk_comm_create (unique id through torch.distributed), ncclBroadcast of X,
        # restarts sharded inside libnmfk_hip, ONE ncclAllGather of the device result buffers per sweep
        try:
            comm = NMFk.parallel.attach(ctx)
            ctx.set_profiling(True)  # (the broadcast of X is timed too: "comm_bcast_X" of nmfk_get_profile)
            comm.bcast_X(X if rank == 0 else None, root=0)
            bcast_prof = ctx.get_profile().get("comm_bcast_X")
            ctx.set_profiling(False)
        except Exception as e:  # fail loudly with the rank: a silent hang in the next collective helps nobody
            print(f"[bench rank {rank}/{world}] joining the RCCL communicator failed: {e!r}", file=sys.stderr, flush=True)
            raise
    elif world > 1:  # CPU-side rehearsal of the sharding logic (gloo): torch.distributed carries the data
        X = np.asfortranarray(NMFk.parallel.broadcast_X(X if rank == 0 else None))
        ctx.set_X(X)
    elif multi is not None:
        ctx.set_profiling(True)
        multi.set_X(X)  # every logical rank's context holds X (nmfk_multi_set_X: the broadcast of the loopback transport)
        bcast_prof = ctx.get_profile().get("comm_bcast_X")
        ctx.set_profiling(False)
    else:
        ctx.set_X(X)  # X resident in HBM (column-major + row-major copies) before the timed region

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(seed):
        try:
            return NMFk.execute(X, ks, args.nruns, load=False, save=False, quiet=True, seed=seed, ctx=ctx,
                                maxiter=args.maxiter, compute=args.compute, return_details=True)
        except Exception as e:
            print(f"[bench rank {rank}/{world}] sweep failed: {e!r}", file=sys.stderr, flush=True)
            raise

    for w in range(args.warmup):
        step(1000 + w)
    if not args.no_profile:
        ctx.set_profiling(True)
    sync()
    t0 = time.perf_counter()
    out = None
    for s in range(args.steps):
        out = step(1 + s)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = ctx.get_profile() if not args.no_profile else {}
    # N > 1: what the communication cost, per rank (nmfk_get_profile "comm_*": host wall time of the steps of nmfk_mu_sweep_sharded /
    # nmfk_comm_bcast on each rank, bytes in the `flops` field), gathered on rank 0
    comm_all = None
    if (world > 1 or multi is not None) and prof:
        mine = {k_: v for k_, v in prof.items() if k_.startswith("comm_")}
        if world > 1:
            comm_all = [None] * world
            dist.all_gather_object(comm_all, mine)
        else:
            comm_all = [mine]
    W, H, fit, rob, aic, kopt, details = out
    nfact = len(ks) * args.nruns
    value = args.steps * nfact / dt

    if rank == 0:
        iters_by_k = {k: details[k]["iters"] for k in ks}
        total_iters = int(sum(int(np.sum(v)) for v in iters_by_k.values()))
        line = {
            "metric": "NMF factorizations/sec (nruns x |krange|), default stop rule, incl. clustering+silhouettes+kopt",
            "value": value, "unit": "factorizations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.compute, "data": "synthetic",
            "config": {"workload": f"dense U(0,1) fp32 X {args.n}x{args.m}, k={args.kmin}:{args.kmax}, nruns={args.nruns} "
                                   f"(BASELINE.json configs[2] / north-star 1-GPU target; {nfact} factorizations per step)",
                       "stop_rule": f"reference defaults: maxiter={args.maxiter}, tol=1e-19, tolOF=1e-3, maxbaditers=10, maxreattempts=2",
                       "parallelism": (f"REHEARSAL: restarts sharded over {args.gpus} LOGICAL ranks on one GPU (loopback transport of "
                                       f"libnmfk_hip, nmfk_multi_*); not a multi-GPU measurement" if multi is not None else
                                       f"restarts sharded over {world} rank(s)" + (" (C ABI: nmfk_comm_* / RCCL)" if comm else "")),
                       "kopt": kopt,
                       "mean_iterations_per_factorization": total_iters / nfact},
        }
        if comm_all:
            def per_sweep(name, f=max):  # ms per sweep (the timed steps), over the ranks
                v = [c_[name]["ms"] / args.steps for c_ in comm_all if name in c_]
                return f(v) if v else None

            ag = [c_["comm_allgather"] for c_ in comm_all if "comm_allgather" in c_]
            lw = [c_["comm_bcast"] for c_ in comm_all if "comm_bcast" in c_]
            line["comm"] = {
                "transport": "RCCL (ncclBroadcast / ncclAllGather behind the C ABI)" if comm is not None else
                             ("loopback (N logical ranks on ONE GPU: device-to-device copies + a host barrier; a rehearsal, no xGMI)" if multi is not None
                              else f"torch.distributed {backend} (host rehearsal)"),
                "rccl_ranks": world if comm is not None else (args.gpus if multi is not None else 0),
                "ranks_reporting": len(comm_all),
                "bcast_X_ms": bcast_prof["ms"] if bcast_prof else None, "bcast_X_bytes": bcast_prof["flops"] if bcast_prof else None,
                "slowest_rank_sweep_s": (per_sweep("comm_local_sweep", max) or 0) / 1e3, "fastest_rank_sweep_s": (per_sweep("comm_local_sweep", min) or 0) / 1e3,
                "wait_for_slowest_rank_ms_per_sweep": per_sweep("comm_wait_for_ranks", max),
                "allgather_ms_per_sweep": per_sweep("comm_allgather", max),
                "allgather_bytes_received_per_rank_and_sweep": (ag[0]["flops"] / args.steps) if ag else None,
                "deliver_ms_per_sweep": per_sweep("comm_deliver", max),
                "lean_W_bcast_ms_per_sweep": per_sweep("comm_bcast", max), "lean_W_bcasts_per_sweep": (lw[0]["launches"] / args.steps) if lw else 0,
                "lean_W_bytes_per_sweep": (lw[0]["flops"] / args.steps) if lw else 0,
                "note": "host wall time of the steps of nmfk_mu_sweep_sharded / nmfk_comm_bcast (libnmfk_hip, nmfk_get_profile 'comm_*'), maximum over "
                        "the ranks unless named otherwise; local sweep = a rank's share of the restarts (no collective inside the MU loop); wait = the "
                        "status agreement behind it, i.e. the time the fastest rank waits for the slowest; all-gather = H, objective, iteration counts of "
                        "every restart (W stays with its owner: best = true); lean W = the winning restart's W from its owner, once per rank k"}
        if prof and "mu_loop" in prof and prof["mu_loop"]["ms"] > 0:
            # Round-3 schedule: every factorization of the sweep (ranks 2..16) runs on the split-operand MFMA half-step, ALL
            # units in one launch per half-step (the kernels switch between their rank variants per workgroup):
            #   H half-step  hyb_step_kernel<2,8,0>   streaming form (loop dimension n = 8192); <2,8,2> behind a check iteration:
            #                the same half-step leaves the monitored objective (deferred check), 1 launch in 10
            #   W half-step  hyb_res_kernel<2,false,false,false>      resident form  (loop factor H, 512 rows, in LDS)
            # A launch has the GPU to itself, so its sampled duration (HIP events on the launching stream, every 47th
            # iteration) is exclusive GPU time; rocprofv3 --kernel-trace --stats of the same command gives the same
            # averages (profiles/r03/bench_default_kernel_stats.csv).
            loop = prof["mu_loop"]
            tf = loop["flops"] / (loop["ms"] * 1e-3) / 1e12
            per_kp = {}
            for name, v in prof.items():
                if name.startswith(("h_step", "w_step")) and v["launches"]:
                    per_kp[name] = {"avg_launch_ms": round(v["ms"] / v["launches"], 4), "sampled_launches": v["launches"],
                                    "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
            dom = max(per_kp, key=lambda k_: prof[k_]["ms"]) if per_kp else None
            xbytes = sum(float(np.sum(iters_by_k[k])) for k in ks) * 2.0 * args.n * args.m * 4.0 * args.steps / max(world, 1)
            whole = {"achieved": tf, "frac": tf / PEAK_FP32_TFLOPS, "mu_loop_gpu_ms": loop["ms"],
                     "note": "algorithmic flops of ALL half-step launches / GPU time of the whole MU loop (HIP events), check blocks included"}
            kernel_of = {"h_step<mfma>": "hyb_step_kernel<2,8,0> (nmfk_step_hyb.hip, streaming form; <2,8,2> = the same with the monitored objective as a by-product behind a check iteration): the H half-step of ALL units of the sweep in one launch",
                         "w_step<mfma>": "hyb_res_kernel<2,false,false,false> (nmfk_step_hyb.hip, resident form: the loop factor H in LDS): the W half-step of ALL units of the sweep in one launch"}
            traffic, tsrc = _pmc_traffic({"h_step<mfma>": "hyb_step_kernel", "w_step<mfma>": "hyb_res_kernel"}.get(dom))
            sched = ctx.last_sweep_info()
            if dom in kernel_of and sched.get("cohorts", 1) > 1:
                # the units run as cohorts on several streams (few-unit sweeps: a rank's share at N > 1): launches overlap, a sampled
                # launch's duration is not exclusive GPU time -- the line's fraction is the whole MU loop's
                line["roofline"] = {
                    "kernel": kernel_of["w_step<mfma>"] + " + " + kernel_of["h_step<mfma>"] + f" -- as {sched['cohorts']} cohorts of units on "
                              "concurrent streams (nmfk_mu_sweep, 'Cohorts')",
                    "bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
                    "traffic": None, "traffic_source": "not attributed: the launches of the cohorts overlap",
                    "whole_mu_loop": whole, "sampled_launches_overlapping": per_kp,
                    "note": "algorithmic flops of all half-step launches (4*n*m*k per half-step per ACTIVE restart) / GPU time of the MU loop; "
                            "the per-launch durations under `sampled_launches_overlapping` include time shared with the other cohort"}
            elif dom in kernel_of:
                d = prof[dom]
                tfd = d["flops"] / (d["ms"] * 1e-3) / 1e12
                other = [k_ for k_ in kernel_of if k_ != dom and k_ in prof and prof[k_]["launches"]]
                line["roofline"] = {
                    "kernel": kernel_of[dom] + f" ({nfact // max(world, 1)} factorizations per launch: kernel variants by rank -- one bf16 MFMA "
                              "+ 4x4x1 fp32 numerator blocks for k <= 4, two + two sets for k <= 8, three + 16-signal fp32 MFMAs for k <= 16)",
                    "bound": "mfma", "achieved": tfd, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tfd / PEAK_FP32_TFLOPS,
                    "traffic": traffic, "traffic_source": tsrc,
                    "avg_launch_ms": d["ms"] / d["launches"], "sampled_launches": d["launches"],
                    "other_half_step": {k_: {"kernel": kernel_of[k_], "avg_launch_ms": prof[k_]["ms"] / prof[k_]["launches"],
                                             "achieved": prof[k_]["flops"] / (prof[k_]["ms"] * 1e-3) / 1e12,
                                             "frac": prof[k_]["flops"] / (prof[k_]["ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS} for k_ in other},
                    "whole_mu_loop": whole,
                    "note": "flops = 4*n*m*k per half-step per ACTIVE restart (W*H and the product with the ratio; SURVEY 8d: "
                            "8*n*m*k per iteration). The kernel computes W*H on the bf16 matrix pipe from exact three-term bf16 "
                            "splits (fp32-accurate) and the numerators on the fp32 matrix pipe; the peak quoted is the fp32 MFMA = "
                            "fp32 vector peak of gfx950 (the bf16 part of the work is cheaper than its fp32 equivalent, so this "
                            "fraction is against the arithmetic the REFERENCE does, not against the pipes' own peaks). "
                            "X is L2/Infinity-Cache resident, HBM is not the bound.",
                    "x_GBps_if_every_restart_streamed_X": xbytes / (loop["ms"] * 1e-3) / 1e9,
                    "x_GBps_note": "SURVEY 8d's 2*n*m*4 B per restart and iteration over the MU-loop time: an ON-DIE figure (L2 / "
                                   "Infinity Cache serve X to the restarts of a launch), NOT HBM traffic and not to be read against "
                                   "the 8 TB/s HBM peak; `traffic` is the measured HBM bytes per launch",
                }
            else:
                line["roofline"] = {
                    "kernel": "step_kernel<KP> / hyb kernels per launch group (schedule other than the bench default)",
                    "bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
                    "traffic": traffic, "traffic_source": tsrc, "mu_loop_gpu_ms": loop["ms"], "dominant_rank_kernel": dom,
                    "per_rank_kernel": per_kp,
                    "note": "flops = 4*n*m*k per half-step per ACTIVE restart; launch groups run concurrently, so the fraction is the "
                            "aggregate over the MU loop.",
                    "x_GBps_if_every_restart_streamed_X": xbytes / (loop["ms"] * 1e-3) / 1e9,
                    "x_GBps_note": "on-die figure (L2 / Infinity Cache), not HBM traffic",
                }
            line["config"]["schedule"] = sched
        # ---- behind the timed region, rank 0, in order of importance, while the run stays inside --budget-s (the driver's run of
        # 20 + 5 steps is ~351 s of sweeps and is cut at 600 s; measured with the default budget of 400 s: 367 s with cfg2 + cfg5, ~383 s
        # with cfg4 too).  What does not fit is listed under
        # `skipped` with the committed line that holds it (profiles/r05/bench_full_line.json: the same command with fewer steps).
        skipped = []

        def fits(name, est_s):
            left = args.budget_s - (time.perf_counter() - t_start)
            if est_s <= left:
                return True
            skipped.append(f"{name} (~{est_s:.0f} s; {left:.0f} s of the {args.budget_s:.0f} s budget left)")
            return False

        if not args.no_cpu_baseline:  # (~1 s)
            threads = min(32, len(os.sched_getaffinity(0)))
            line["cpu_baseline"] = cpu_baseline(X, ks, args.nruns, iters_by_k, threads)
            line["cpu_baseline"]["reference_julia"] = julia_reference_baseline(args.n, args.m, ks, args.nruns)
        default_cfg = world == 1 and multi is None and (args.n, args.m, args.kmin, args.kmax, args.nruns) == (8192, 512, 2, 16, 32)
        sweep_s = dt / args.steps
        k0 = 6

        def planted_sweep(Xp, nruns=args.nruns, **kw):
            t = time.perf_counter()
            o = NMFk.execute(Xp, ks, nruns, load=False, save=False, quiet=True, seed=2, ctx=ctx, return_details=True, **kw)
            sec = time.perf_counter() - t
            its = np.concatenate([o[6][k]["iters"] for k in ks]).astype(np.float64)
            return dict(kopt=o[5], seconds=sec, factorizations_per_s=len(ks) * nruns / sec, mean_iterations=float(its.mean()),
                        min_iterations=int(its.min()), max_iterations=int(its.max()),
                        active_unit_fraction=float(its.sum() / (its.max() * len(its))), schedule=ctx.last_sweep_info(),
                        robustness=[float(v) for v in o[3][ks[0] - 1:]])

        if not args.no_kopt_check and default_cfg:
            ctx.set_profiling(False)
            # (1) "same kopt" half of the metric: SURVEY 8d's planted rank-6 matrix through the same sweep in fp32 (the product)
            if fits("config.kopt_planted: planted rank-6 sweep, fp32", 1.1 * sweep_s + 2):
                Xp = planted_matrix(ctx, args.n, args.m, k0)
                ctx.set_X(Xp)
                p32 = planted_sweep(Xp)
                line["config"]["kopt_planted"] = p32["kopt"]
                line["config"]["kopt_planted_expected"] = k0
                line["config"]["planted"] = {
                    "matrix": f"X = W0 H0 + 0.01 U (rank {k0}, {args.n}x{args.m}; SURVEY 8d), k={args.kmin}:{args.kmax}, nruns={args.nruns}, "
                              "default stop rule, whole execute() incl. clustering",
                    **{k_: p32[k_] for k_ in ("kopt", "seconds", "factorizations_per_s", "mean_iterations", "min_iterations", "max_iterations",
                                               "active_unit_fraction", "robustness")}}
                ctx.set_X(X)
        if not args.no_secondary and world == 1 and multi is None:
            # (2) the other single-GPU BASELINE workloads: driver-visible numbers
            sec = {}
            for name, fn, est in (("cfg2", lambda N_, c_: secondary_cfg2(N_, c_, X), 2), ("cfg4", secondary_cfg4, 12), ("cfg5", secondary_cfg5, 7)):
                if not fits(f"secondary.{name}", est):
                    sec[name] = {"skipped": "time budget (--budget-s); see profiles/r05/bench_full_line.json"}
                    continue
                try:
                    sec[name] = fn(NMFk, ctx)
                except Exception as e:  # noqa: BLE001  (the headline line must not be lost to a secondary measurement)
                    sec[name] = {"error": repr(e)}
            line["secondary"] = sec
            ctx.set_X(X)
        if not args.no_kopt_check and default_cfg:
            ctx.set_profiling(False)
            # (3) kopt of the HEADLINE matrix in fp64 compute (the reference's arithmetic and stop decisions, oracle-verified by
            # tests/test_gpu_parity.py::test_stop_rule_fp64_identical_iterations), 8 restarts per rank: does the Float64 loop see
            # the same kopt on pure noise as the fp32 product?  
            # -- with 8 restarts when they fit, else with 4 (the driver's run of 20 + 5 steps leaves ~15 s here)
            for nr64, est64 in ((8, 26), (4, 14)):
                if fits(f"config.kopt_f64_mode: the headline matrix in fp64 compute, {nr64} restarts", est64):
                    h64 = planted_sweep(X, compute="f64", nruns=nr64)
                    line["config"]["kopt_f64_mode"] = {"kopt": h64["kopt"], "nruns": nr64, "seconds": h64["seconds"], "robustness": h64["robustness"],
                                                       "robustness_f32_product": [float(v) for v in rob[ks[0] - 1:]]}
                    break
            # (4) the planted matrix in fp64 compute (8 restarts: the packed-VALU fp64 kernels take 63 s for all 32; the full-size
            # agreement of the two modes is tests/test_gpu_fullsize.py::test_planted_rank6_same_kopt_at_metric_size)
            if "planted" in line["config"] and fits("config.kopt_planted_f64_mode", 26):
                Xp = planted_matrix(ctx, args.n, args.m, k0)
                ctx.set_X(Xp)
                p64 = planted_sweep(Xp, compute="f64", nruns=8)
                line["config"]["kopt_planted_f64_mode"] = p64["kopt"]
                line["config"]["planted"]["f64_mode"] = {"nruns": 8, "kopt": p64["kopt"], "seconds": p64["seconds"], "robustness": p64["robustness"]}
            # (5) structured data on which the restarts RETIRE: the stop rule's tolOF = 1e-3 is an absolute improvement of the sum of
            # squares (Mult:24, 81), so the planted matrix scaled to entries of ~0.15 stagnates between 1 000 and 10 000 iterations.
            # Retire-aware schedule (default) against the static one (NMFK_REPLAN=0).
            if "planted" in line["config"] and fits("config.planted.retiring", 18):
                Xs = planted_matrix(ctx, args.n, args.m, k0, scale=0.1)
                ctx.set_X(Xs)
                ps = planted_sweep(Xs)
                os.environ["NMFK_REPLAN"] = "0"
                try:
                    ps0 = planted_sweep(Xs)
                finally:
                    del os.environ["NMFK_REPLAN"]
                line["config"]["planted"]["retiring"] = {
                    "matrix": "X = 0.1 * (W0 H0 + 0.01 U): the same matrix scaled so that the ABSOLUTE tolOF = 1e-3 of the "
                              "reference's stop rule (Mult:24, 81) retires restarts between 1 000 and 10 000 iterations",
                    **ps, "static_schedule_seconds": ps0["seconds"], "static_schedule_kopt": ps0["kopt"],
                    "speedup_of_the_retire_aware_schedule": ps0["seconds"] / ps["seconds"]}
            ctx.set_X(X)
        if skipped:
            line["skipped"] = {"what": skipped, "why": f"--budget-s {args.budget_s:.0f}: the whole run stays inside it",
                               "see": "profiles/r05/bench_full_line.json (the same command with --steps 3 --warmup 1: every entry present)"}
        line["wall_s_total"] = time.perf_counter() - t_start
        print(json.dumps(line))
    if multi is not None:
        multi.close()
    if world > 1:
        dist.barrier()
        if comm is not None:
            NMFk.parallel.detach(ctx)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
