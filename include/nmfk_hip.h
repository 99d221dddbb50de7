/*
 * nmfk_hip.h -- C ABI of libnmfk_hip.so: the MI355X (gfx950) implementation of the NMFk.jl
 * `execute(X, krange, nNMF; method=:simple)` hot path.
 *
 * The reference (pure Julia, /root/reference) has no FFI for this path; the seam is the Julia function
 * signature.  Each entry point below names the reference function(s) it replaces (file:line relative to
 * the reference root) -- a Julia maintainer binds them with `ccall` (INTEGRATION.md shows the stub), the
 * tests bind them with ctypes.
 *
 * Conventions
 *  - All matrices are COLUMN-MAJOR (Julia native), float32 on the boundary (T = Float32, the BASELINE dtype).
 *  - Every bulk pointer may be a host pointer or a HIP device pointer; the library copies with
 *    hipMemcpyDefault into/out of its own HBM workspace and never retains a caller pointer after return.
 *  - Every function returns an nmfk_status (0 = ok).  No C++ exception crosses the boundary;
 *    nmfk_last_error() gives the message of the last failure on the calling thread.
 *    (Exception: nmfk_set_X_csc builds its CSR twin on the host and takes HOST pointers.)
 *  - A context is bound to one GPU and is not re-entrant.  Multi-GPU = one context per GPU, as one process per GPU
 *    (nmfk_comm_*: the host layer passes the RCCL unique id around) or as threads of one process (nmfk_multi_*, what a
 *    Julia caller uses); the (k, restart) work list is sharded by restart inside the library ("multi-GPU" below).
 */
#ifndef NMFK_HIP_H
#define NMFK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmfk_ctx nmfk_ctx;

typedef enum {
  NMFK_OK = 0,
  NMFK_ERR_BAD_ARG = 1,      /* null pointer, zero dimension (Exec:242-244), size mismatch (Mult:40,50)       */
  NMFK_ERR_NEGATIVE = 2,     /* "All matrix entries must be nonnegative!" (Mult:4-7)                          */
  NMFK_ERR_NAN_INIT = 3,     /* "Initial values for the W/H matrix entries include NaNs!" (Mult:42-44,52-54)  */
  NMFK_ERR_NO_X = 4,         /* nmfk_set_X has not been called                                                */
  NMFK_ERR_HIP = 5,          /* a HIP runtime call failed                                                      */
  NMFK_ERR_UNSUPPORTED = 6,  /* k > NMFK_MAX_K                                                                 */
  NMFK_ERR_NO_DEVICE = 7,    /* no usable gfx950 device: the library has NO CPU fallback                      */
  NMFK_ERR_RCCL = 8          /* librccl missing, or an RCCL call failed (the message names the call and the rank) */
} nmfk_status;

enum { NMFK_MAX_K = 64 };

/* why a restart left the MU loop (Mult:64, 75-78, 112-115) */
enum { NMFK_STOP_MAXITER = 1, NMFK_STOP_STAGNATION = 2, NMFK_STOP_TOL = 3, NMFK_STOP_CONSISTENCY = 4 };

/* arithmetic of the inner loop */
enum {
  NMFK_COMPUTE_F32 = 0, /* W, H, W*H in fp32; column/row sums and the objective accumulated in fp64 (default) */
  NMFK_COMPUTE_F64 = 1  /* everything in fp64, like the reference's default path (Mult:38,48): validation    */
};

/* keyword arguments of NMFmultiplicative (Mult:24) as forwarded by execute_singlerun_compute (Exec:729,762) */
typedef struct {
  double tol;            /* 1e-19  (Exec:729)  objective < tol => stop                                        */
  double tolOF;          /* 1e-3   (Mult:24)   absolute SSE improvement per 10 iterations                    */
  double lambda;         /* 1e-32  (Mult:24)   replacement of zeros / first-iteration value of missing data  */
  double weight;         /* 1      (Mult:24)   scalar weight of the monitored objective (Mult:74)            */
  int64_t maxiter;       /* 10000  (Exec:729)                                                                 */
  int32_t maxreattempts; /* 2      (Mult:24)                                                                  */
  int32_t maxbaditers;   /* 10     (Mult:24)                                                                  */
  int32_t stopconv;      /* 1000   (Mult:24)                                                                  */
  int32_t Wfixed;        /* 0      (Mult:24,69)                                                               */
  int32_t Hfixed;        /* 0      (Mult:24,66)                                                               */
  int32_t normalize;     /* 1      modifymatrices (Exec:486-489, 795-805): rows of H sum to 1, W rescaled    */
  int32_t compute;       /* NMFK_COMPUTE_F32                                                                  */
  int32_t reserved;
} nmfk_mu_params;

/* library / device ---------------------------------------------------------------------------------------- */
int nmfk_version(void);
const char *nmfk_last_error(void);
int nmfk_device_count(int *count);
/* Creates a context on GPU `device`.  Fails with NMFK_ERR_NO_DEVICE when there is none. */
int nmfk_create(int device, nmfk_ctx **out);
int nmfk_destroy(nmfk_ctx *ctx);
int nmfk_device_info(nmfk_ctx *ctx, char *name, int name_len, int *compute_units, int64_t *hbm_bytes);
/* Fills *p with the reference defaults listed above. */
int nmfk_mu_default_params(nmfk_mu_params *p);

/* data ---------------------------------------------------------------------------------------------------- */
/* Replaces NMFpreprocessing! (Mult:3-22) + the implicit capture of X by execute_singlerun (Exec:535-541).
 * X: n x m, leading dimension ldx >= n.  NaN = missing.  Entries <= 0 become lambda in the device copy (the
 * caller's array is never modified, so nothing has to be restored as in Mult:123-124).  Keeps two HBM
 * copies: column-major (W half-step, objective) and row-major (H half-step).
 * nan_count / zero_count (optional) receive count(isnan, X) and count(X .<= 0). */
int nmfk_set_X(nmfk_ctx *ctx, const float *X, int64_t n, int64_t m, int64_t ldx, double lambda, int64_t *nan_count,
               int64_t *zero_count);

/* Sparse X in CSC form (BASELINE configs[3], "zeros stay zeros").  The reference turns every zero into
 * lambda = 1e-32 (Mult:17-18), so only the stored non-zeros contribute to the update ratios; the gather kernels
 * selected by this call are that arithmetic to < 1e-30 and stream 8 B + 4k B per non-zero instead of the dense
 * n x m passes.  colptr: m+1 offsets, rowidx/vals: nnz entries (entries <= 0 are dropped, negative => error,
 * NaN (missing data) needs the dense path).  *kept (optional) receives the number of stored non-zeros.
 * Row indices need NOT be ascending inside a column (the library sorts a column that is not; Julia's SparseMatrixCSC
 * always is); duplicate (row, column) entries are kept as separate records whose contributions add.
 * Also builds, on the host, the CSR twin and the sliced-ELL copies of both orientations that the blocked form of the
 * half-steps reads (ranks up to 32; skipped for an orientation whose padding would exceed 4 slots per non-zero).
 * n, m <= 2^24 (NMFK_ERR_UNSUPPORTED beyond: the sparse kernels index a factor's elements with 32 bits). */
int nmfk_set_X_csc(nmfk_ctx *ctx, int64_t n, int64_t m, int64_t nnz, const int64_t *colptr, const int32_t *rowidx,
                   const float *vals, int64_t *kept);

/* Array-valued `weight` of the monitored objective sum((((X - W*H) .* weight)[.!inan]).^2) (Mult:74,125; the
 * assertion on its shape is Exec:484).  weight: n x m column-major (the host layer broadcasts vector forms), or NULL
 * to clear.  It multiplies the scalar nmfk_mu_params.weight.  Cleared by nmfk_set_X. */
int nmfk_set_weight(nmfk_ctx *ctx, const float *weight, int64_t n, int64_t m);

/* Portable counter-based U(0,1) generator standing in for Julia's rand (Mult:38,48); bit-identical to the
 * oracle's (oracle/nmfk_oracle.c).  out[i] = u(seed, offset + i), i < count. */
int nmfk_fill_uniform(nmfk_ctx *ctx, uint64_t seed, uint64_t offset, int64_t count, float *out);

/* multiplicative updates ------------------------------------------------------------------------------------ */
/* Replaces the restart loop of execute_run (Exec:527-543) around execute_singlerun_compute(:simple)
 * (Exec:729-807) around NMFmultiplicative (Mult:24-127), for ALL ranks of a sweep at once (the reference's
 * `for nk in nkrange` at Exec:203 is serial; here every (k, restart) pair is one unit of a flat work list).
 *
 *  nk, ks[nk]        ranks to factorize (1 <= k <= NMFK_MAX_K)
 *  nruns             restarts per rank
 *  Winit[q], Hinit[q]  q < nk: nruns stacked n x ks[q] / ks[q] x m initial factors, or NULL (also the arrays
 *                    themselves may be NULL) => generated on the device from seeds (W first, then H, Mult:38,48)
 *  seeds[q*nruns+r]  seed of restart r of rank ks[q]; required where inits are NULL
 *  W_out[q]          nruns stacked n x ks[q]   (after the Exec:801-803 normalisation when params->normalize)
 *  H_out[q]          nruns stacked ks[q] x m
 *  frob_out[q]       nruns: objvalue = normnan(X - W*H) (Exec:791-792), rounded to T like objvalue::Vector{T}
 *  sse_out[q]        nruns: sum(((X-W*H).*weight)[.!inan].^2) (Mult:125)            (optional: NULL)
 *  iters_out[q]      nruns: iterations executed                                      (optional)
 *  reason_out[q]     nruns: NMFK_STOP_*                                              (optional)
 */
int nmfk_mu_sweep(nmfk_ctx *ctx, int nk, const int32_t *ks, int nruns, const float *const *Winit,
                  const float *const *Hinit, const uint64_t *seeds, const nmfk_mu_params *params, float *const *W_out,
                  float *const *H_out, float *const *frob_out, double *const *sse_out, int32_t *const *iters_out,
                  int32_t *const *reason_out);

/* One rank: the body of execute_run's restart loop (Exec:527-543).  Same arguments, without the outer arrays. */
int nmfk_mu_batch(nmfk_ctx *ctx, int k, int nruns, const float *Winit, const float *Hinit, const uint64_t *seeds,
                  const nmfk_mu_params *params, float *W_out, float *H_out, float *frob_out, double *sse_out,
                  int32_t *iters_out, int32_t *reason_out);

/* robustness ------------------------------------------------------------------------------------------------ */
/* Replaces clustersolutions(HBig[idxsort][idxsol], false) (Clus:425-517, call site Exec:623) and the
 * silhouette part of finalize(W, H, idx, false) (Fin:36-66, call site Exec:637), in T = Float32 as the
 * reference does for Float32 X (Clus:463).
 *  Hstack     nsol stacked k x m solutions, ALREADY sorted by objective and filtered by the caller
 *  labels     k x nsol, 1-based cluster of signal a of solution t (labels[:,1] = 1:k)
 *  centroids  k x m   (running sums / nsol, Clus:512-516)
 *  point_sil  k x nsol silhouettes (NaN -> 0, Fin:58)
 *  cluster_sil k      mean silhouette per cluster (Fin:66); robustness = minimum (Exec:638) is the caller's */
int nmfk_cluster_silhouette(nmfk_ctx *ctx, int k, int nsol, int64_t m, const float *Hstack, int32_t *labels,
                            float *centroids, float *point_sil, float *cluster_sil);

/* Silhouettes for GIVEN labels (the second half of nmfk_cluster_silhouette on its own).  Needed by the
 * clusterWmatrix=true path, where the reference clusters the W matrices in place -- the first solution's W becomes
 * the centroid (Clus:453-455, 484, 512) -- and `finalize` (Fin:45-50) then computes the silhouettes on the MUTATED
 * stack with the labels found before.  stack: nsol x (k x len) signal-major, labels k x nsol. */
int nmfk_silhouette(nmfk_ctx *ctx, int k, int nsol, int64_t m, const float *stack, const int32_t *labels,
                    float *point_sil, float *cluster_sil);

/* Cluster means and corrected variances of W and H (Fin:64-77; used when best=false, Exec:655-658).
 *  Wstack nsol x (n x k), Hstack nsol x (k x m), labels k x nsol  ->  Wmean, Wvar (n x k), Hmean, Hvar (k x m) */
int nmfk_cluster_stats(nmfk_ctx *ctx, int k, int nsol, int64_t n, int64_t m, const float *Wstack, const float *Hstack,
                       const int32_t *labels, float *Wmean, float *Hmean, float *Wvar, float *Hvar);

/* normnan(X - W*H) with NaN -> skipped (Help:226-228): the re-checks at Exec:603, 664-667 and 212-222. */
int nmfk_frobenius(nmfk_ctx *ctx, int k, const float *W, const float *H, double *out);

/* robust k-means ---------------------------------------------------------------------------------------------- */
/* robustkmeans(X, k, repeats; maxiter, tol, compute_silhouettes_flag)  src/NMFkCluster.jl:172-246,
 * without the JLD cache: `repeats` independent Clustering.kmeans(X, k; distance=CosineDist()) runs (k-means++
 * seeding, Lloyd iterations; restated in oracle/nmfk_oracle.c), all concurrent on the GPU; the run with the lowest
 * total cost wins (first wins ties, Clus:227) and its clusters are relabelled by decreasing size (sortclustering,
 * Clus:264-292).  Random draws: the library's counter-based generator, u(seed + repeat, draw index).
 *   X            d x n column-major, columns = samples (the reference's orientation)
 *   assignments  n, 1-based, sorted labels;  centers d x k (sorted order, zero columns for clusters not found)
 *   costs        n cosine distances to the own centre;  counts k (sorted);  totalcost = sum(costs)
 *   nclusters    clusters found (< k: the reference warns, Clus:232-234)
 *   all_costs    repeats total costs (may be NULL);  silhouettes n point silhouettes of the best run on
 *                pairwise(CosineDist(), zerostoepsilon(X)) (Clus:204-213), NULL = compute_silhouettes_flag=false
 *   converged    KmeansResult.converged of the winning repeat (1: the centre movement fell below tol before maxiter;
 *                may be NULL) */
int nmfk_robustkmeans_ex(nmfk_ctx *ctx, int d, int64_t n, const float *X, int k, int repeats, int maxiter, double tol,
                         uint64_t seed, int32_t *assignments, float *centers, float *costs, int32_t *counts,
                         double *totalcost, int32_t *best_repeat, int32_t *iterations, int32_t *nclusters,
                         double *all_costs, float *silhouettes, int32_t *converged);
/* The same without `converged`: the signature this entry point had before the flag existed (nmfk_version() 200).  An exported symbol
 * keeps its argument list -- a caller built against the earlier header must not make the library write through an argument it never
 * passed -- so the flag came with a new name (as nmfk_last_sweep_info_ex did); nmfk_version() is 210 since. */
int nmfk_robustkmeans(nmfk_ctx *ctx, int d, int64_t n, const float *X, int k, int repeats, int maxiter, double tol,
                      uint64_t seed, int32_t *assignments, float *centers, float *costs, int32_t *counts,
                      double *totalcost, int32_t *best_repeat, int32_t *iterations, int32_t *nclusters,
                      double *all_costs, float *silhouettes);

/* multi-GPU -------------------------------------------------------------------------------------------------- */
/* Replaces the reference's only parallelism on this path, Distributed.pmap over the restarts of one rank
 * (src/NMFkExecute.jl:511-526, X shipped to a worker with every task): rank g of N owns the restarts {g, g+N, ...} of
 * EVERY rank k of the sweep; X is broadcast once (ncclBroadcast, device to device over xGMI), nothing is exchanged
 * inside the MU loop, and the results of all restarts are exchanged in ONE ncclAllGather of equally sized device
 * buffers, so that every rank can run the robustness step.  librccl is loaded at run time (dlopen).            */
typedef struct nmfk_comm nmfk_comm;
enum { NMFK_UNIQUE_ID_BYTES = 128 };
/* restarts of one rank k that shard `rank` of `nranks` runs: *count real ones, padded to *padded = ceil(nruns/nranks)
 * (short lists repeat their last restart; the padding results are dropped).  Pure host arithmetic. */
int nmfk_shard_plan(int nruns, int nranks, int rank, int32_t *count, int32_t *padded);
/* the inverse: restart r (0-based) of every rank k is run by shard *rank as its local restart *slot (r = rank + slot * nranks). */
int nmfk_shard_owner(int nruns, int nranks, int r, int32_t *rank, int32_t *slot);
/* ncclGetUniqueId: called by ONE rank, the host layer hands the 128 bytes to the others (Julia: Distributed / a file;
 * bench.py: torch.distributed).  nmfk_comm_create is collective over the nranks contexts (ncclCommInitRank). */
int nmfk_comm_unique_id(void *id128);
int nmfk_comm_create(nmfk_ctx *ctx, int nranks, int rank, const void *id128, nmfk_comm **out);
int nmfk_comm_destroy(nmfk_comm *comm);
int nmfk_comm_info(nmfk_comm *comm, int *rank, int *nranks);
/* Collective.  X (n x m, leading dimension ldx; host or device) is read on `root` only -- the other ranks pass NULL and
 * learn n, m from the broadcast; every rank then runs NMFpreprocessing! (nmfk_set_X) on its device copy. */
int nmfk_comm_bcast_X(nmfk_comm *comm, int root, const float *X, int64_t n, int64_t m, int64_t ldx, double lambda,
                      int64_t *n_out, int64_t *m_out, int64_t *nan_count, int64_t *zero_count);
/* Collective.  `bytes` bytes at `buf` (host or device) of rank `root` arrive at `buf` of every other rank.  The host layer uses it
 * for the lean result exchange of execute_run with best = true (Exec:655-658; SURVEY 8e: "send of the winning W"): after
 * nmfk_mu_sweep_sharded(need_W = 0) the owner of the restart with the lowest objective broadcasts its W (n x k floats). */
int nmfk_comm_bcast(nmfk_comm *comm, int root, void *buf, int64_t bytes);
/* Collective nmfk_mu_sweep: identical arguments on every rank, describing ALL nruns restarts (seeds, optional inits,
 * outputs).  Each rank runs its shard and receives the results of every restart; pass H_out = NULL on a rank that
 * does not need them.  need_W = 0: the W matrices are not exchanged (W_out then receives this rank's own restarts
 * only; the best=true, clusterWmatrix=false path of execute_run needs the W of one restart per rank k). */
int nmfk_mu_sweep_sharded(nmfk_ctx *ctx, nmfk_comm *comm, int nk, const int32_t *ks, int nruns,
                          const float *const *Winit, const float *const *Hinit, const uint64_t *seeds,
                          const nmfk_mu_params *params, int need_W, float *const *W_out, float *const *H_out,
                          float *const *frob_out, double *const *sse_out, int32_t *const *iters_out,
                          int32_t *const *reason_out);
/* Error behaviour of the collective calls: before every data collective the ranks agree on a status word, so when the
 * local step of ONE rank fails (bad X on the root, out of memory, a NaN initial factor in its shard: NMFK_ERR_NAN_INIT, ...)
 * EVERY rank returns that status -- the failing rank with its own message, the others naming the rank -- and no rank is
 * left blocked in a collective (the reference's pmap rethrows a worker's exception on the caller, Exec:511-526). */

/* Loopback transport -- a TEST HOOK that executes the N > 1 code of this section on a one-GPU box: `nranks` logical ranks
 * = `nranks` contexts on ONE GPU driven by `nranks` host threads of one process; the collectives are a host barrier plus
 * device-to-device copies instead of RCCL, everything else (shard plan, padding, contribution layout, strided delivery,
 * status agreement, thread fan-out of nmfk_multi_*) is the code the RCCL transport runs. */
typedef struct nmfk_loop_group nmfk_loop_group;
int nmfk_loopback_group_create(int nranks, nmfk_loop_group **out);
int nmfk_loopback_group_destroy(nmfk_loop_group *group);
int nmfk_comm_create_loopback(nmfk_ctx *ctx, nmfk_loop_group *group, int rank, nmfk_comm **out);

/* One process, several GPUs (what `NMFkHIP.execute(...; ngpus = 8)` calls): GPUs 0..ngpus-1, one context, communicator
 * and host thread each; the results are delivered through GPU 0.  nmfk_multi_context gives GPU g's context (GPU 0:
 * clustering, silhouettes and fit re-checks after the sweep).  Sparse X: nmfk_set_X_csc on every GPU's context (it takes
 * host pointers; nmfk_multi_set_X broadcasts dense X only), then nmfk_multi_sweep as usual. */
typedef struct nmfk_multi nmfk_multi;
int nmfk_multi_create(int ngpus, nmfk_multi **out);
int nmfk_multi_destroy(nmfk_multi *mh);
/* the loopback form of nmfk_multi_create (test hook): nranks logical ranks on GPU `device` */
int nmfk_multi_create_loopback(int nranks, int device, nmfk_multi **out);
/* rank `gpu`'s communicator (not owned by the caller): per-rank calls of nmfk_mu_sweep_sharded from the caller's own threads */
int nmfk_multi_comm(nmfk_multi *mh, int gpu, nmfk_comm **comm);
int nmfk_multi_context(nmfk_multi *mh, int gpu, nmfk_ctx **ctx);
int nmfk_multi_set_X(nmfk_multi *mh, const float *X, int64_t n, int64_t m, int64_t ldx, double lambda, int64_t *nan_count,
                     int64_t *zero_count);
int nmfk_multi_sweep(nmfk_multi *mh, int nk, const int32_t *ks, int nruns, const float *const *Winit,
                     const float *const *Hinit, const uint64_t *seeds, const nmfk_mu_params *params, float *const *W_out,
                     float *const *H_out, float *const *frob_out, double *const *sse_out, int32_t *const *iters_out,
                     int32_t *const *reason_out);

/* measurement ----------------------------------------------------------------------------------------------- */
/* HIP-event timing of the MU kernels on the streams they are launched on (bench.py's roofline leg).
 * After a sweep with profiling enabled: names[i] / total_ms[i] / launches[i] / flops[i] for i < *count:
 *   "mu_loop"                     GPU wall time of the whole MU loop (all rank groups run concurrently) and the
 *                                 algorithmic flops of every half-step in it (4*n*m*k per ACTIVE restart)
 *   "h_step<kp>" / "w_step<kp>"   sampled launches (every 47th iteration) of the half-step kernel of one rank,
 *                                 each timed on its own stream, with the flops of the restarts active in them
 *                                 ("<mfma>": the mixed-rank launch group on the matrix-pipe kernels; with cohorts -- NMFK_COHORTS --
 *                                 launches of different cohorts overlap: the durations are then not exclusive GPU time)
 *   "comm_*"                      multi-GPU calls (nmfk_comm_bcast_X, nmfk_mu_sweep_sharded, nmfk_comm_bcast): HOST wall time of
 *                                 their steps on this rank -- comm_bcast_X (ncclBroadcast of X to completion), comm_local_sweep
 *                                 (this rank's share), comm_wait_for_ranks (the status agreement behind it: returns when the slowest
 *                                 rank's sweep has ended), comm_allgather, comm_deliver (strided copies into the caller's arrays),
 *                                 comm_bcast (the winning restart's W) -- with the BYTES moved in the `flops` field */
int nmfk_set_profiling(nmfk_ctx *ctx, int enabled);
/* Launch schedule the LAST nmfk_mu_sweep on this context chose (tests assert that the schedule they mean to cover was
 * the one taken; bench.py reports it).  info[0] = phases of the sweep (2 = the split-operand MFMA group first, the
 * packed-VALU ranks afterwards), info[1] = units on the split-operand MFMA half-step, info[2] = mixed-rank packed-VALU
 * launch groups, info[3] = launch groups in all, info[4] = units on the all-MFMA half-step (k > 16); the retire-aware schedule
 * (one launch group on the matrix-pipe kernels; NMFK_REPLAN=0 switches it off): info[5] = re-plans executed, info[6] = tier
 * of the last plan (tier j is planned for ceil(units / 2^j) units), info[7] = units in the work list of the last plan. */
int nmfk_last_sweep_info(nmfk_ctx *ctx, int32_t info[8]);
/* The same with `count` <= 16 entries: info[8] = checks (per launch group) whose objective came out of the following H half-step
 * (the deferred check, see NMFK_DEFER_OBJ below), info[9] = checks with an objective launch of their own, info[10] = cohorts of the
 * matrix-pipe launch group (its units dealt to that many streams, see NMFK_COHORTS below; 1 = one launch per half-step), info[11] = H
 * half-step launches whose partial numerators were summed by the W half-step behind them instead of a reduce launch (NMFK_FUSE_RED); the rest 0. */
int nmfk_last_sweep_info_ex(nmfk_ctx *ctx, int32_t *info, int count);
/* Test hook, pure host arithmetic (no device): the tiers of the retire-aware schedule for a sweep of `units` units of ranks 2..16
 * (widest kernel variant 4 / 8 / 16) in one launch group on the matrix-pipe kernels, on a GPU of `cus` CUs -- tier j is the launch
 * geometry for ceil(units / 2^j) units; the sweep switches to it when the units still active fit.  Row j of `out` (16 ints per
 * row, <= cap rows): units; then for the H and for the W half-step: workgroups per unit of the resident form (0 = streaming form),
 * wsplit, S, dchunk, fused, table slots, slots a unit writes; then the cohorts such a group would run as.  *count = rows written.
 * variant 0: the ranks 2..16 in equal numbers (8 : 4 : 3 units of the variants 16 : 8 : 4, the bench sweep's mix). */
int nmfk_plan_hyb_tiers(int64_t n, int64_t m, int variant, int units, int cus, int32_t *out, int cap, int *count);
/* The objective the stop rule monitors -- sum((((X - W*H) .* weight)[.!inan]).^2) every 10th iteration (Mult:73-74) -- as
 * the device computed it, for every check of every restart of the NEXT sweeps (parity tests compare it with the oracle's
 * trace check by check).  nmfk_get_objective_trace: restart `restart` of rank ks[kidx] of the last sweep; *count = checks made. */
int nmfk_set_objective_trace(nmfk_ctx *ctx, int enabled);
int nmfk_get_objective_trace(nmfk_ctx *ctx, int kidx, int restart, double *out, int cap, int *count);
int nmfk_get_profile(nmfk_ctx *ctx, int max_entries, char (*names)[64], double *total_ms, int64_t *launches,
                     double *flops, int *count);

/* Environment variables the library reads (at the start of every sweep; tests and A/B measurements -- the defaults are the
 * product and need none of them):
 *   NMFK_HYB          0 / 1: split-operand MFMA half-step off / on for the ranks >= NMFK_HYB_MINK (default: on for every rank 2..16 of a
 *                     dense fp32 sweep without missing data, except a few units of ranks <= 4 only)
 *   NMFK_HYB_PHASES   0 / 1: the matrix-pipe ranks as ONE launch group that runs first, the other ranks behind it (default: 1)
 *   NMFK_HYB_RES      0: no resident form of that half-step (short loop dimension: the loop factor in LDS)
 *   NMFK_MERGE        g: the ranks <= 16 share g mixed-rank packed-VALU launch groups (default: by restarts per rank)
 *   NMFK_MFMA_WIDE    0: ranks > 16 on the packed-VALU kernel;  NMFK_WIDE2 0: on the all-fp32 MFMA kernel only;
 *   NMFK_MFMA_SSE     0: their monitored objective on the packed-VALU objective kernel
 *   NMFK_REPLAN       0: static launch schedule (no re-planning as restarts retire); 2: re-plan at every tier (tests)
 *   NMFK_CLAMP_ALWAYS 1: the clamp pass of a check block (Mult:99-100) scans every unit (default: only units whose half-step kernels
 *                     wrote a value below eps() in the check iteration; the same elements are clamped either way, the results agree to rounding)
 *   NMFK_DEFER_OBJ    0: every check block computes its monitored objective (Mult:74) in a launch of its own (default on the
 *                     matrix-pipe kernels and for sparse X in the blocked form: the H half-step that follows a check iteration leaves it as a by-product and the check's
 *                     tests run behind that half-step; same stop decisions, the objective taken after the clamp instead of before:
 *                     <= 1e-13 of its value)
 *   NMFK_WIDE_GROUPS  0: a launch group per rank above 16; f: the ranks of one kernel instantiation (32 / 48 / 64 signals) share launch
 *                     groups of up to f workgroups per CU (default 2)
 *   NMFK_COHORTS      c: the launch group on the matrix-pipe kernels (ranks 2..16) runs as c cohorts of units, each on its own stream, so
 *                     that one cohort's half-step fills the CUs another's leaves idle (default: by the group's size; same bits per unit)
 *   NMFK_FUSE_RED     1 (EXPERIMENTAL, off by default): an H half-step whose loop range is split over workgroups gets no reduce launch when the
 *                     W half-step behind it runs the resident form -- that launch sums the partial numerators while it stages H.  The new H has
 *                     reduce_kernel's bits; rowsum(H) is added in another order, so the results differ from the default path by rounding.
 *                     Every workgroup of a unit repeats the sum: measured 2.5 % slower on a 60-unit share (profiles/r05/dense_probes.txt)
 *   NMFK_HYB_LAG      0 / 1: the matrix-pipe streaming half-step never / always runs its second lane tile one chunk late (default: where a wave
 *                     walks 32 chunks or more; same bits either way).  Exception: the half-step that also leaves the deferred check's objective
 *                     (1 launch in 10) always runs the lagged form -- its monitored objective is compared with the plain check order
 *                     (NMFK_DEFER_OBJ=0) by tests/test_gpu_parity.py::test_deferred_check_against_the_plain_order
 *   NMFK_WIDE_BN      ranks 17..64 (wide2_step_kernel): 0 = numerators on the fp32 matrix pipe (rounds 3-5), 1 (default) = on the bf16 pipe from exact
 *                     three-term splits of the ratios where the padded width is 48 or 64 signals, 2 = at 32 signals too (slower there)
 *   NMFK_SP_BLK       0: sparse X in the gather form only (no sliced-ELL copies are built); 2: blocked form whatever the size
 *   NMFK_TARGET_WGS   workgroups a half-step launch should have before loop ranges are split (default 2 x CUs)
 *   NMFK_STREAMS      concurrent launch-group streams (8; 16 for a dozen launch groups or more; sparse X: 1);  NMFK_HOST_TIMING=1: host / GPU wait times on stderr
 *   NMFK_RCCL_LIB     the RCCL shared object nmfk_comm_* loads (default: librccl.so.1 ...) */

#ifdef __cplusplus
}
#endif
#endif
