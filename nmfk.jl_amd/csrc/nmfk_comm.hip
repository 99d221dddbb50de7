// libnmfk_hip: multi-GPU layer of the C ABI (include/nmfk_hip.h, "multi-GPU").
//
// The reference's only parallelism on this path is `Distributed.pmap` over the restarts of ONE rank
// (src/NMFkExecute.jl:511-526), shipping X to a worker with every task.  Here the (k, restart) units of the WHOLE sweep
// are sharded: rank g owns the restarts {g, g + N, ...} of every k (the cost of a unit grows with k, so every rank
// gets the same mix of ranks), X is broadcast ONCE, nothing is exchanged inside the MU loop, and the results of all
// restarts are exchanged in ONE all-gather of equally sized device buffers, so that every rank can run the (tiny)
// robustness step.  RCCL (ncclBroadcast / ncclAllGather over xGMI) moves device buffers; no host staging.
//
// librccl is loaded at run time (dlopen): a single-GPU user needs no RCCL, and inside a PyTorch process the copy that
// is already loaded is the one used.  One rank = one nmfk_ctx = one GPU; ranks may be processes (unique id passed by
// the host layer) or threads of one process (nmfk_multi_*).
//
// No rank ever enters a data collective alone: before each one the ranks AGREE on a status word (a 4-byte all-gather of
// the local return code), so a rank whose local step failed (bad X on the root, out of memory, a NaN initial factor in
// its shard, ...) makes EVERY rank return an error instead of leaving the others blocked in the collective.
//
// Transport: RCCL, or -- a test hook -- a LOOPBACK group: N logical ranks as N host threads with N contexts on ONE GPU,
// collectives emulated by a host barrier plus device-to-device copies (nmfk_multi_create_loopback).  Everything above
// the three collective primitives (shard plan, padding, contribution layout, strided delivery, thread fan-out, status
// agreement) is the same code either way, which is how the N > 1 path is executed on a one-GPU test box.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "nmfk_ctx.h"

namespace {

// the few RCCL entry points used (signatures of rccl.h; the types are opaque here)
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[NMFK_UNIQUE_ID_BYTES];
} ncclUniqueId;
enum { ncclSuccess = 0, ncclInt8 = 0, ncclChar = 0, ncclFloat32 = 7 };
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
};
Rccl &rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    // NMFK_RCCL_LIB names the copy to use (a process must hold ONE librccl: inside PyTorch, the one it ships)
    const char *names[] = {getenv("NMFK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names)
      if (nm && *nm && (R.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!R.h) {
      R.err = std::string("librccl could not be loaded (") + (dlerror() ? dlerror() : "?") + ")";
      return;
    }
    auto sym = [&](const char *s) {
      void *p = dlsym(R.h, s);
      if (!p && R.err.empty()) R.err = std::string("librccl lacks ") + s;
      return p;
    };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.Broadcast = (decltype(R.Broadcast))sym("ncclBroadcast");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  });
  return R;
}

}  // namespace

// loopback group: a reusable host barrier and the per-rank buffer addresses of the collective in flight
struct nmfk_loop_group {
  int n = 1;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  std::vector<const void *> ptr;  // [n] source buffer of each rank
  std::vector<int32_t> word;      // [n] status words
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++arrived == n) {
      arrived = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

struct nmfk_comm {
  nmfk_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;        // RCCL transport ...
  nmfk_loop_group *loop = nullptr;  // ... or the loopback group (not owned)
  int nranks = 1, rank = 0;
  DevBuf send, recv, xbuf, stat;
};

namespace {

int rccl_fail(const nmfk_comm *c, const char *what, int rc) {
  char b[512];
  snprintf(b, sizeof(b), "RCCL error %d (%s) in %s on rank %d of %d", rc,
           rccl().GetErrorString ? rccl().GetErrorString(rc) : "?", what, c ? c->rank : -1, c ? c->nranks : -1);
  return fail(NMFK_ERR_RCCL, b);
}
#define RCCLCHECK(c, what, expr)                   \
  do {                                             \
    const int _r = (expr);                         \
    if (_r != ncclSuccess) return rccl_fail(c, what, _r); \
  } while (0)

// ---- the three collective primitives (device buffers, the context's stream) ---------------------------------------
// broadcast `bytes` of `buf` from `root` (in place)
int coll_bcast(nmfk_comm *c, void *buf, size_t bytes, int root, const char *what) {
  hipStream_t st = c->ctx->stream;
  if (!c->loop) {
    RCCLCHECK(c, what, rccl().Broadcast(buf, buf, bytes, ncclChar, root, c->comm, st));
    return NMFK_OK;
  }
  nmfk_loop_group *G = c->loop;
  HIPCHECK(hipStreamSynchronize(st));  // the root's buffer is complete
  G->ptr[c->rank] = buf;
  G->barrier();
  hipError_t e = hipSuccess;
  if (c->rank != root) {
    e = hipMemcpyAsync(buf, G->ptr[root], bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  G->barrier();  // nobody touches the root's buffer before every rank has its copy
  HIPCHECK(e);
  return NMFK_OK;
}
// all-gather: `bytes` from every rank's `send` into recv[h * bytes], h = rank of origin
int coll_allgather(nmfk_comm *c, const void *send, void *recv, size_t bytes, const char *what) {
  hipStream_t st = c->ctx->stream;
  if (!c->loop) {
    RCCLCHECK(c, what, rccl().AllGather(send, recv, bytes, ncclChar, c->comm, st));
    return NMFK_OK;
  }
  nmfk_loop_group *G = c->loop;
  HIPCHECK(hipStreamSynchronize(st));
  G->ptr[c->rank] = send;
  G->barrier();
  hipError_t e = hipSuccess;
  for (int h = 0; h < c->nranks && e == hipSuccess; ++h)
    e = hipMemcpyAsync((char *)recv + (size_t)h * bytes, G->ptr[h], bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  G->barrier();
  HIPCHECK(e);
  return NMFK_OK;
}
// what a rank's status word holds between agreements (0xEE bytes): see agree()
constexpr int NMFK_STATUS_POISON_BYTE = 0xEE;
constexpr int32_t NMFK_STATUS_POISON = (int32_t)0xEEEEEEEEu;
// Status agreement: every rank contributes the return code of its local step; all ranks get NMFK_OK, or the first
// failing rank's code.  A rank that failed itself keeps its own message (nmfk_last_error); the others name the rank.
int agree(nmfk_comm *c, int my_rc, const char *step) {
  std::vector<int32_t> all((size_t)c->nranks, NMFK_OK);
  if (c->loop) {
    nmfk_loop_group *G = c->loop;
    G->word[c->rank] = my_rc;
    G->barrier();
    for (int h = 0; h < c->nranks; ++h) all[h] = G->word[h];
    G->barrier();
  } else {
    hipStream_t st = c->ctx->stream;
    int32_t *d = (int32_t *)c->stat.p;  // [0]: mine, [64 ...]: everybody's (allocated by nmfk_comm_create)
    const int32_t mine = my_rc;
    // A local HIP failure must not keep this rank out of the all-gather (the peers would wait in it for ever): between
    // agreements d[0] holds NMFK_STATUS_POISON, so a status word that could not be uploaded reaches the peers as a failure.
    if (hipMemcpyAsync(d, &mine, sizeof(mine), hipMemcpyHostToDevice, st) != hipSuccess && my_rc == NMFK_OK)
      my_rc = fail(NMFK_ERR_HIP, std::string("could not upload the status word of ") + step);
    RCCLCHECK(c, "ncclAllGather(status)", rccl().AllGather(d, d + 64, sizeof(int32_t), ncclChar, c->comm, st));
    hipError_t e = hipMemcpyAsync(all.data(), d + 64, sizeof(int32_t) * c->nranks, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipMemsetAsync(d, NMFK_STATUS_POISON_BYTE, sizeof(int32_t), st);
    if (e != hipSuccess && my_rc == NMFK_OK)  // (every rank has contributed; the peers' words are unknown to this rank)
      my_rc = fail(NMFK_ERR_HIP, std::string("could not read the status words of ") + step);
  }
  if (my_rc != NMFK_OK) return my_rc;  // (message of the local failure stays)
  for (int h = 0; h < c->nranks; ++h)
    if (all[h] != NMFK_OK) {
      char b[256];
      const bool poison = all[h] == NMFK_STATUS_POISON;
      snprintf(b, sizeof(b), "rank %d of %d failed with status %d in %s%s (this is rank %d; see that rank's nmfk_last_error)", h,
               c->nranks, poison ? (int)NMFK_ERR_HIP : (int)all[h], step, poison ? " (it could not upload its status word)" : "",
               c->rank);
      return fail(poison ? NMFK_ERR_HIP : all[h], b);
    }
  return NMFK_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Host wall time of a step of a collective call into the context's profile (nmfk_get_profile, with nmfk_set_profiling on):
// entries "comm_*", total_ms / launches as for the kernels, the `flops` field holds the BYTES the step moved.  Round 5 (VERDICT r4
// item 6): a first run on more than one GPU should explain itself -- how long the broadcast of X, the wait for the slowest rank,
// the all-gather and the delivery took next to the local sweep.
struct CommTimer {
  nmfk_ctx *ctx;
  const char *name;
  double bytes;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  CommTimer(nmfk_ctx *c, const char *n, double b = 0) : ctx(c), name(n), bytes(b) {}
  void stop() {
    if (!ctx || !ctx->profiling || !name) return;
    auto &E = ctx->prof[name];
    E.ms += 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    E.launches += 1;
    E.flops += bytes;
    name = nullptr;
  }
  ~CommTimer() { stop(); }
};

}  // namespace

#define NMFK_EXPORT extern "C" __attribute__((visibility("default")))

NMFK_EXPORT int nmfk_shard_plan(int nruns, int nranks, int rank, int32_t *count, int32_t *padded) {
  if (nruns <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(NMFK_ERR_BAD_ARG, "bad shard arguments");
  if (count) *count = rank < nruns ? (nruns - rank + nranks - 1) / nranks : 0;  // restarts {rank, rank + N, ...}
  if (padded) *padded = (nruns + nranks - 1) / nranks;                          // every rank runs this many (short lists repeat their last restart)
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_comm_unique_id(void *id128) {
  if (!id128) return fail(NMFK_ERR_BAD_ARG, "id is null");
  Rccl &R = rccl();
  if (!R.err.empty()) return fail(NMFK_ERR_RCCL, R.err);
  ncclUniqueId id;
  const int rc = R.GetUniqueId(&id);
  if (rc != ncclSuccess) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
  memcpy(id128, &id, NMFK_UNIQUE_ID_BYTES);
  return NMFK_OK;
}

namespace {
int comm_new(nmfk_ctx *ctx, int nranks, int rank, nmfk_comm **out) {
  nmfk_comm *c = new nmfk_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  if (c->stat.ensure(sizeof(int32_t) * (64 + (size_t)nranks))) {  // status words of agree(): never allocated on a failure path
    delete c;
    return fail(NMFK_ERR_HIP, "out of device memory (communicator)");
  }
  (void)hipMemset(c->stat.p, NMFK_STATUS_POISON_BYTE, sizeof(int32_t));  // (agree(): the word between agreements)
  *out = c;
  return NMFK_OK;
}
}  // namespace

NMFK_EXPORT int nmfk_comm_create(nmfk_ctx *ctx, int nranks, int rank, const void *id128, nmfk_comm **out) {
  if (!ctx || !id128 || !out) return fail(NMFK_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NMFK_ERR_BAD_ARG, "bad rank / nranks");
  Rccl &R = rccl();
  if (!R.err.empty()) return fail(NMFK_ERR_RCCL, R.err);
  HIPCHECK(hipSetDevice(ctx->device));
  nmfk_comm *c = nullptr;
  const int rc0 = comm_new(ctx, nranks, rank, &c);
  if (rc0 != NMFK_OK) return rc0;
  ncclUniqueId id;
  memcpy(&id, id128, NMFK_UNIQUE_ID_BYTES);
  const int rc = R.CommInitRank(&c->comm, nranks, id, rank);
  if (rc != ncclSuccess) {
    const int e = rccl_fail(c, "ncclCommInitRank", rc);
    c->stat.release();
    delete c;
    return e;
  }
  *out = c;
  return NMFK_OK;
}

// Loopback transport (test hook, see the head of this file): the `nranks` communicators of one group belong to `nranks`
// contexts on ONE GPU and are driven by `nranks` host threads of one process.
NMFK_EXPORT int nmfk_loopback_group_create(int nranks, nmfk_loop_group **out) {
  if (!out || nranks < 1) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  nmfk_loop_group *G = new nmfk_loop_group();
  G->n = nranks;
  G->ptr.assign((size_t)nranks, nullptr);
  G->word.assign((size_t)nranks, 0);
  *out = G;
  return NMFK_OK;
}
NMFK_EXPORT int nmfk_loopback_group_destroy(nmfk_loop_group *G) {
  delete G;
  return NMFK_OK;
}
NMFK_EXPORT int nmfk_comm_create_loopback(nmfk_ctx *ctx, nmfk_loop_group *group, int rank, nmfk_comm **out) {
  if (!ctx || !group || !out) return fail(NMFK_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  if (rank < 0 || rank >= group->n) return fail(NMFK_ERR_BAD_ARG, "bad rank");
  HIPCHECK(hipSetDevice(ctx->device));
  nmfk_comm *c = nullptr;
  const int rc0 = comm_new(ctx, group->n, rank, &c);
  if (rc0 != NMFK_OK) return rc0;
  c->loop = group;
  *out = c;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_comm_destroy(nmfk_comm *c) {
  if (!c) return NMFK_OK;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  c->send.release();
  c->recv.release();
  c->xbuf.release();
  c->stat.release();
  delete c;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_comm_info(nmfk_comm *c, int *rank, int *nranks) {
  if (!c) return fail(NMFK_ERR_BAD_ARG, "comm is null");
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_comm_bcast_X(nmfk_comm *c, int root, const float *X, int64_t n, int64_t m, int64_t ldx, double lambda,
                                  int64_t *n_out, int64_t *m_out, int64_t *nan_count, int64_t *zero_count) {
  if (!c) return fail(NMFK_ERR_BAD_ARG, "comm is null");
  nmfk_ctx *ctx = c->ctx;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // step 1 (local): arguments; the size of X is known on the root only
  auto local_args = [&]() -> int {
    if (root < 0 || root >= c->nranks) return fail(NMFK_ERR_BAD_ARG, "bad root");
    if (c->rank == root && (!X || n <= 0 || m <= 0 || ldx < n))
      return fail(NMFK_ERR_BAD_ARG, n <= 0 || m <= 0 ? "Input array has a zero dimension!" : "root: bad X");
    if (c->xbuf.ensure(256)) return fail(NMFK_ERR_HIP, "out of device memory");
    return NMFK_OK;
  };
  int rc = agree(c, local_args(), "nmfk_comm_bcast_X (arguments)");
  if (rc != NMFK_OK) return rc;
  int64_t hdr[2] = {c->rank == root ? n : 0, c->rank == root ? m : 0};
  HIPCHECK(hipMemcpyAsync(c->xbuf.p, hdr, sizeof(hdr), hipMemcpyHostToDevice, st));
  rc = coll_bcast(c, c->xbuf.p, sizeof(hdr), root, "ncclBroadcast(size of X)");
  if (rc != NMFK_OK) return rc;
  HIPCHECK(hipMemcpyAsync(hdr, c->xbuf.p, sizeof(hdr), hipMemcpyDeviceToHost, st));
  HIPCHECK(hipStreamSynchronize(st));
  n = hdr[0];
  m = hdr[1];
  // step 2 (local): the broadcast buffer and, on the root, the dense n x m image of the caller's (host or device) array
  const size_t bytes = (size_t)n * (size_t)m * sizeof(float);
  auto local_stage = [&]() -> int {
    if (n <= 0 || m <= 0) return fail(NMFK_ERR_BAD_ARG, "Input array has a zero dimension!");
    if (c->xbuf.ensure(bytes)) return fail(NMFK_ERR_HIP, "out of device memory (X broadcast buffer)");
    if (c->rank == root)
      HIPCHECK(hipMemcpy2DAsync(c->xbuf.p, (size_t)n * 4, X, (size_t)ldx * 4, (size_t)n * 4, (size_t)m, hipMemcpyDefault, st));
    return NMFK_OK;
  };
  rc = agree(c, local_stage(), "nmfk_comm_bcast_X (staging)");
  if (rc != NMFK_OK) return rc;
  {
    CommTimer tm(ctx, "comm_bcast_X", (double)bytes);  // (enqueue to completion on this rank's stream)
    rc = coll_bcast(c, c->xbuf.p, bytes, root, "ncclBroadcast(X)");
    if (rc != NMFK_OK) return rc;
    HIPCHECK(hipStreamSynchronize(st));
  }
  // step 3 (local): NMFpreprocessing! on every rank; a rank that fails here (negative entries fail on all) fails them all
  rc = agree(c, nmfk_set_X(ctx, (const float *)c->xbuf.p, n, m, n, lambda, nan_count, zero_count), "nmfk_set_X");
  if (n_out) *n_out = n;
  if (m_out) *m_out = m;
  return rc;
}

// Collective: `bytes` bytes at `buf` (host or device memory) of rank `root` arrive in every other rank's `buf`.  The lean result
// exchange of execute_run with best = true (Exec:655-658): after nmfk_mu_sweep_sharded with need_W = 0 every rank knows the
// objective of every restart, and the owner of the best one hands its W (n x k floats) to the others -- SURVEY 8e's "send of the
// winning W" -- instead of every W travelling.
NMFK_EXPORT int nmfk_comm_bcast(nmfk_comm *c, int root, void *buf, int64_t bytes) {
  if (!c) return fail(NMFK_ERR_BAD_ARG, "comm is null");
  HIPCHECK(hipSetDevice(c->ctx->device));
  hipStream_t st = c->ctx->stream;
  auto local_stage = [&]() -> int {
    if (root < 0 || root >= c->nranks || bytes < 0 || (bytes > 0 && !buf)) return fail(NMFK_ERR_BAD_ARG, "bad broadcast arguments");
    if (bytes == 0) return NMFK_OK;
    if (c->xbuf.ensure((size_t)bytes)) return fail(NMFK_ERR_HIP, "out of device memory (broadcast buffer)");
    if (c->rank == root) HIPCHECK(hipMemcpyAsync(c->xbuf.p, buf, (size_t)bytes, hipMemcpyDefault, st));
    return NMFK_OK;
  };
  CommTimer tm(c->ctx, "comm_bcast", (double)bytes);  // (staging, status agreement, broadcast, copy out)
  int rc = agree(c, local_stage(), "nmfk_comm_bcast (staging)");
  if (rc != NMFK_OK || bytes == 0) return rc;
  rc = coll_bcast(c, c->xbuf.p, (size_t)bytes, root, "ncclBroadcast(bytes)");
  if (rc != NMFK_OK) return rc;
  if (c->rank != root) HIPCHECK(hipMemcpyAsync(buf, c->xbuf.p, (size_t)bytes, hipMemcpyDefault, st));
  HIPCHECK(hipStreamSynchronize(st));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_shard_owner(int nruns, int nranks, int r, int32_t *rank, int32_t *slot) {
  if (nruns <= 0 || nranks <= 0 || r < 0 || r >= nruns) return fail(NMFK_ERR_BAD_ARG, "bad shard arguments");
  if (rank) *rank = r % nranks;  // restart r = rank + slot * nranks
  if (slot) *slot = r / nranks;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_mu_sweep_sharded(nmfk_ctx *ctx, nmfk_comm *c, int nk, const int32_t *ks, int nruns,
                                      const float *const *Winit, const float *const *Hinit, const uint64_t *seeds,
                                      const nmfk_mu_params *params, int need_W, float *const *W_out, float *const *H_out,
                                      float *const *frob_out, double *const *sse_out, int32_t *const *iters_out,
                                      int32_t *const *reason_out) {
  if (!ctx || !c || c->ctx != ctx) return fail(NMFK_ERR_BAD_ARG, "null argument / communicator of another context");
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int N = c->nranks, g = c->rank;
  const int64_t n = ctx->n, m = ctx->m;
  int32_t mine = 0, cpad = 0;
  const bool withW = need_W != 0;
  // layout of a rank's contribution: per rank k (in the caller's order) cpad restarts of W, H, frob, iters, reason, sse
  std::vector<size_t> oW, oH, oF, oI, oR, oS, oWl, oWi, oHi;
  size_t tot = 0;
  // step 1 (local): arguments and buffers.  Every way out of a local step leads to agree(): no rank is left alone in a collective.
  auto local_plan = [&]() -> int {
    if (!ks || !params) return fail(NMFK_ERR_BAD_ARG, "null argument");
    if (nk <= 0 || nruns <= 0) return fail(NMFK_ERR_BAD_ARG, "nk and nruns must be positive");
    if (!ctx->Xc && !ctx->sparse) return fail(NMFK_ERR_NO_X, "nmfk_set_X / nmfk_comm_bcast_X has not been called");
    (void)nmfk_shard_plan(nruns, N, g, &mine, &cpad);
    oW.assign(nk, 0), oH.assign(nk, 0), oF.assign(nk, 0), oI.assign(nk, 0), oR.assign(nk, 0), oS.assign(nk, 0);
    oWl.assign(nk, 0), oWi.assign(nk, 0), oHi.assign(nk, 0);
    for (int q = 0; q < nk; ++q) {
      if (ks[q] < 1 || ks[q] > NMFK_MAX_K) return fail(NMFK_ERR_UNSUPPORTED, "k out of range");
      const size_t k = (size_t)ks[q];
      oW[q] = tot;
      tot += align256(withW ? sizeof(float) * cpad * k * n : 0);
      oH[q] = tot;
      tot += align256(sizeof(float) * cpad * k * m);
      oF[q] = tot;
      tot += align256(sizeof(float) * cpad);
      oI[q] = tot;
      tot += align256(sizeof(int32_t) * cpad);
      oR[q] = tot;
      tot += align256(sizeof(int32_t) * cpad);
      oS[q] = tot;
      tot += align256(sizeof(double) * cpad);
    }
    // local scratch behind the contribution: W of the local restarts when it is not exchanged, local inits
    size_t loc = tot;
    for (int q = 0; q < nk; ++q) {
      const size_t k = (size_t)ks[q];
      if (!withW) {
        oWl[q] = loc;
        loc += align256(sizeof(float) * cpad * k * n);
      }
      if (Winit && Winit[q]) {
        oWi[q] = loc;
        loc += align256(sizeof(float) * cpad * k * n);
      }
      if (Hinit && Hinit[q]) {
        oHi[q] = loc;
        loc += align256(sizeof(float) * cpad * k * m);
      }
    }
    if (c->send.ensure(loc)) return fail(NMFK_ERR_HIP, "out of device memory (shard buffers)");
    if (c->recv.ensure(tot * (size_t)N)) return fail(NMFK_ERR_HIP, "out of device memory (gather buffer)");
    return NMFK_OK;
  };
  int rc = agree(c, local_plan(), "nmfk_mu_sweep_sharded (arguments, buffers)");
  if (rc != NMFK_OK) return rc;
  char *S = c->send.p;

  // step 2 (local): this rank's restarts r = g + j*N (j < mine), padded to cpad by repeating the last one
  auto local_sweep = [&]() -> int {
    if (mine <= 0) return NMFK_OK;
    std::vector<uint64_t> lseeds((size_t)nk * cpad, 0);
    std::vector<const float *> wi(nk, nullptr), hi(nk, nullptr);
    std::vector<float *> wo(nk), ho(nk), fo(nk);
    std::vector<double *> so(nk);
    std::vector<int32_t *> io(nk), ro(nk);
    for (int q = 0; q < nk; ++q) {
      const size_t k = (size_t)ks[q];
      for (int j = 0; j < cpad; ++j) {
        const int r = g + std::min(j, mine - 1) * N;
        if (seeds) lseeds[(size_t)q * cpad + j] = seeds[(size_t)q * nruns + r];
        if (Winit && Winit[q])
          HIPCHECK(hipMemcpyAsync(S + oWi[q] + sizeof(float) * j * k * n, Winit[q] + (size_t)r * k * n, sizeof(float) * k * n,
                                  hipMemcpyDefault, st));
        if (Hinit && Hinit[q])
          HIPCHECK(hipMemcpyAsync(S + oHi[q] + sizeof(float) * j * k * m, Hinit[q] + (size_t)r * k * m, sizeof(float) * k * m,
                                  hipMemcpyDefault, st));
      }
      wi[q] = (Winit && Winit[q]) ? (const float *)(S + oWi[q]) : nullptr;
      hi[q] = (Hinit && Hinit[q]) ? (const float *)(S + oHi[q]) : nullptr;
      wo[q] = (float *)(S + (withW ? oW[q] : oWl[q]));
      ho[q] = (float *)(S + oH[q]);
      fo[q] = (float *)(S + oF[q]);
      so[q] = (double *)(S + oS[q]);
      io[q] = (int32_t *)(S + oI[q]);
      ro[q] = (int32_t *)(S + oR[q]);
    }
    HIPCHECK(hipStreamSynchronize(st));
    // the local sweep writes straight into the contribution (device pointers on the boundary)
    return nmfk_mu_sweep(ctx, nk, ks, cpad, wi.data(), hi.data(), seeds ? lseeds.data() : nullptr, params, wo.data(), ho.data(),
                         fo.data(), so.data(), io.data(), ro.data());
  };
  {
    CommTimer tl(ctx, "comm_local_sweep");  // this rank's share: nmfk_mu_sweep of its restarts
    const int lrc = local_sweep();
    tl.stop();
    CommTimer tw(ctx, "comm_wait_for_ranks");  // the status agreement behind it: returns when the SLOWEST rank's sweep has ended
    rc = agree(c, lrc, "its local sweep");
  }
  if (rc != NMFK_OK) return rc;
  {
    CommTimer tg(ctx, "comm_allgather", (double)tot * N);  // (bytes received; enqueue to completion on this rank's stream)
    rc = coll_allgather(c, S, c->recv.p, tot, "ncclAllGather(results)");
    if (rc != NMFK_OK) return rc;
    // (profiling only: the wait for the collective is booked under comm_allgather instead of under the first copy of comm_deliver, which sits
    //  behind it on the same stream and would wait for it anyway -- the call's total does not change, only where the wait is accounted)
    if (ctx->profiling) HIPCHECK(hipStreamSynchronize(st));
  }
  CommTimer td(ctx, "comm_deliver");  // strided copies into the caller's arrays

  // deliver: restart r = h + j*N of rank k comes from rank h, slot j
  if (H_out) {
    for (int h = 0; h < N; ++h) {
      int32_t cnt = 0;
      (void)nmfk_shard_plan(nruns, N, h, &cnt, nullptr);
      if (cnt == 0) continue;
      const char *Rb = c->recv.p + (size_t)h * tot;
      for (int q = 0; q < nk; ++q) {
        const size_t k = (size_t)ks[q];
        auto scatter = [&](void *dst0, const char *src, size_t elem) -> hipError_t {  // cnt pieces, destination stride N pieces
          if (!dst0) return hipSuccess;
          return hipMemcpy2DAsync((char *)dst0 + elem * h, elem * N, src, elem, elem, (size_t)cnt, hipMemcpyDefault, st);
        };
        if (withW && W_out && W_out[q]) HIPCHECK(scatter(W_out[q], Rb + oW[q], sizeof(float) * k * n));
        if (H_out[q]) HIPCHECK(scatter(H_out[q], Rb + oH[q], sizeof(float) * k * m));
        if (frob_out && frob_out[q]) HIPCHECK(scatter(frob_out[q], Rb + oF[q], sizeof(float)));
        if (iters_out && iters_out[q]) HIPCHECK(scatter(iters_out[q], Rb + oI[q], sizeof(int32_t)));
        if (reason_out && reason_out[q]) HIPCHECK(scatter(reason_out[q], Rb + oR[q], sizeof(int32_t)));
        if (sse_out && sse_out[q]) HIPCHECK(scatter(sse_out[q], Rb + oS[q], sizeof(double)));
      }
    }
    // W of this rank's own restarts when W is not exchanged (the caller's stack then holds W for these restarts only)
    if (!withW && W_out && mine > 0)
      for (int q = 0; q < nk; ++q)
        if (W_out[q]) {
          const size_t e = sizeof(float) * (size_t)ks[q] * n;
          HIPCHECK(hipMemcpy2DAsync((char *)W_out[q] + e * g, e * N, S + oWl[q], e, e, (size_t)mine, hipMemcpyDefault, st));
        }
  }
  HIPCHECK(hipStreamSynchronize(st));
  return NMFK_OK;
}

// ------------------------------------------------------------------------------------------------------
// one process, several GPUs: a context, a communicator and a host thread per GPU
// ------------------------------------------------------------------------------------------------------
struct nmfk_multi {
  std::vector<nmfk_ctx *> ctx;
  std::vector<nmfk_comm *> comm;
  nmfk_loop_group *loop = nullptr;  // loopback form: N logical ranks on one GPU (owned)
};

namespace {
template <class F>
int on_all(nmfk_multi *mh, F f) {
  const int N = (int)mh->ctx.size();
  std::vector<int> rc(N, NMFK_OK);
  std::vector<std::string> msg(N);
  std::vector<std::thread> th;
  for (int g = 0; g < N; ++g)
    th.emplace_back([&, g] {
      rc[g] = f(g);
      if (rc[g] != NMFK_OK) msg[g] = nmfk_last_error();
    });
  for (auto &t : th) t.join();
  // report the rank whose own step failed (the others only say "rank h failed", see agree())
  for (int pass = 0; pass < 2; ++pass)
    for (int g = 0; g < N; ++g)
      if (rc[g] != NMFK_OK && (pass == 1 || msg[g].find("see that rank's nmfk_last_error") == std::string::npos))
        return fail(rc[g], "GPU " + std::to_string(g) + ": " + msg[g]);
  return NMFK_OK;
}
}  // namespace

NMFK_EXPORT int nmfk_multi_destroy(nmfk_multi *mh) {
  if (!mh) return NMFK_OK;
  for (auto *c : mh->comm) (void)nmfk_comm_destroy(c);
  for (auto *x : mh->ctx) (void)nmfk_destroy(x);
  delete mh->loop;
  delete mh;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_multi_create(int ngpus, nmfk_multi **out) {
  if (!out || ngpus < 1) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  *out = nullptr;
  int have = 0;
  (void)nmfk_device_count(&have);
  if (have < ngpus) return fail(NMFK_ERR_NO_DEVICE, "fewer GPUs visible than requested");
  nmfk_multi *mh = new nmfk_multi();
  mh->ctx.assign(ngpus, nullptr);
  mh->comm.assign(ngpus, nullptr);
  char id[NMFK_UNIQUE_ID_BYTES];
  int rc = nmfk_comm_unique_id(id);
  for (int g = 0; g < ngpus && rc == NMFK_OK; ++g) rc = nmfk_create(g, &mh->ctx[g]);
  if (rc == NMFK_OK) rc = on_all(mh, [&](int g) { return nmfk_comm_create(mh->ctx[g], ngpus, g, id, &mh->comm[g]); });
  if (rc != NMFK_OK) {
    const std::string keep = nmfk_last_error();
    (void)nmfk_multi_destroy(mh);
    return fail(rc, keep);
  }
  *out = mh;
  return NMFK_OK;
}

// test hook: `nranks` logical ranks (contexts, communicators, host threads) on the ONE GPU `device`, loopback transport
NMFK_EXPORT int nmfk_multi_create_loopback(int nranks, int device, nmfk_multi **out) {
  if (!out || nranks < 1) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  *out = nullptr;
  nmfk_multi *mh = new nmfk_multi();
  mh->ctx.assign(nranks, nullptr);
  mh->comm.assign(nranks, nullptr);
  int rc = nmfk_loopback_group_create(nranks, &mh->loop);
  for (int g = 0; g < nranks && rc == NMFK_OK; ++g) rc = nmfk_create(device, &mh->ctx[g]);
  for (int g = 0; g < nranks && rc == NMFK_OK; ++g) rc = nmfk_comm_create_loopback(mh->ctx[g], mh->loop, g, &mh->comm[g]);
  if (rc != NMFK_OK) {
    const std::string keep = nmfk_last_error();
    (void)nmfk_multi_destroy(mh);
    return fail(rc, keep);
  }
  *out = mh;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_multi_comm(nmfk_multi *mh, int gpu, nmfk_comm **comm) {
  if (!mh || !comm || gpu < 0 || gpu >= (int)mh->comm.size()) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  *comm = mh->comm[gpu];
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_multi_set_X(nmfk_multi *mh, const float *X, int64_t n, int64_t m, int64_t ldx, double lambda,
                                 int64_t *nan_count, int64_t *zero_count) {
  if (!mh || !X) return fail(NMFK_ERR_BAD_ARG, "null argument");
  return on_all(mh, [&](int g) {
    return nmfk_comm_bcast_X(mh->comm[g], 0, g == 0 ? X : nullptr, n, m, ldx, lambda, nullptr, nullptr, g == 0 ? nan_count : nullptr,
                             g == 0 ? zero_count : nullptr);
  });
}

NMFK_EXPORT int nmfk_multi_context(nmfk_multi *mh, int gpu, nmfk_ctx **ctx) {
  if (!mh || !ctx || gpu < 0 || gpu >= (int)mh->ctx.size()) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  *ctx = mh->ctx[gpu];
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_multi_sweep(nmfk_multi *mh, int nk, const int32_t *ks, int nruns, const float *const *Winit,
                                 const float *const *Hinit, const uint64_t *seeds, const nmfk_mu_params *params,
                                 float *const *W_out, float *const *H_out, float *const *frob_out, double *const *sse_out,
                                 int32_t *const *iters_out, int32_t *const *reason_out) {
  if (!mh) return fail(NMFK_ERR_BAD_ARG, "null argument");
  return on_all(mh, [&](int g) {  // GPU 0 delivers the results; the others only contribute
    const bool d = g == 0;
    return nmfk_mu_sweep_sharded(mh->ctx[g], mh->comm[g], nk, ks, nruns, Winit, Hinit, seeds, params, 1, d ? W_out : nullptr,
                                 d ? H_out : nullptr, d ? frob_out : nullptr, d ? sse_out : nullptr, d ? iters_out : nullptr,
                                 d ? reason_out : nullptr);
  });
}
