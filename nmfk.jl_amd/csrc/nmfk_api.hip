// libnmfk_hip: C ABI (include/nmfk_hip.h) and host orchestration of the MU sweep.
//
// Host-side structure of one sweep (replaces Exec:203 `for nk in nkrange` x Exec:535-541 `for i = 1:nNMF`):
// every (k, restart) pair is one unit of a flat list sorted by k descending; all units advance in
// lock-step, one kernel launch per half-step covers all of them (grid.y = unit), and a unit whose stop rule
// fired simply stops contributing workgroups.  Per iteration: H numerators, H finish, W numerators, W finish;
// every 10th iteration: objective partials + the check block (Mult:73-117).  The host never waits for the
// GPU inside the loop: the unit states are copied to pinned memory after each check and inspected one check
// later, so the queue stays full.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/nmfk_hip.h"
#include "nmfk_common.h"

#include "nmfk_ctx.h"

namespace {

void free_sparse(nmfk_ctx *ctx) {
  void *ps[] = {ctx->colptr, ctx->rowptr, ctx->rec_csc, ctx->rec_csr, ctx->ell[0], ctx->ell[1], ctx->ellptr[0], ctx->ellptr[1]};
  ctx->ell[0] = ctx->ell[1] = nullptr;
  ctx->ellptr[0] = ctx->ellptr[1] = nullptr;
  ctx->ell_ngb[0] = ctx->ell_ngb[1] = 0;
  for (void *q : ps)
    if (q) (void)hipFree(q);
  ctx->colptr = ctx->rowptr = nullptr;
  ctx->rec_csc = ctx->rec_csr = nullptr;
  ctx->sparse = false;
  ctx->nnz = 0;
}

struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

int ensure_pinned(nmfk_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->pinned_cap) return 0;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  ctx->pinned = nullptr;
  ctx->pinned_cap = 0;
  if (hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault) != hipSuccess) return 1;
  ctx->pinned_cap = bytes;
  return 0;
}

// HIP-event timing of sampled half-step launches, each on the stream it runs on.
struct Sample {
  int kind, group, it;
  size_t e0, e1;
  int u0 = 0, cnt = 0, epoch = 0;  // the launch's units as positions of the work list of re-plan `epoch`
};
struct Sampler {
  nmfk_ctx *ctx;
  std::vector<Sample> samples;
  size_t used = 0;
  bool on;
  explicit Sampler(nmfk_ctx *c) : ctx(c), on(c->profiling) {}
  size_t event() {
    if (used == ctx->events.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) {
        on = false;
        return 0;
      }
      ctx->events.push_back(e);
    }
    return used++;
  }
  // sparse sampling (the events cost host time too); the period is coprime to the 10-iteration check cadence so that
  // the samples cover every phase of it (every 50th iteration always follows a check block: biased short)
  bool want(int it) const { return on && (it % 47) == 23; }
  size_t begin(hipStream_t s) {
    const size_t e = event();
    if (on) (void)hipEventRecord(ctx->events[e], s);
    return e;
  }
  void end(size_t e0, int kind, int group, int it, hipStream_t s, int u0, int cnt, int epoch) {
    const size_t e1 = event();
    if (!on) return;
    (void)hipEventRecord(ctx->events[e1], s);
    samples.push_back({kind, group, it, e0, e1, u0, cnt, epoch});
  }
};

// Schedule overrides for tests and A/B measurements (the defaults are the product; documented in include/nmfk_hip.h):
// environment variables, read in this ONE place at the start of every sweep (a sweep runs seconds; tests flip them between
// sweeps).
//   NMFK_TARGET_WGS   workgroups a half-step launch should have before loop ranges are split (default 2 x CUs)
//   NMFK_MFMA_WIDE    0: ranks > 16 on the packed-VALU kernel instead of the all-MFMA one
//   NMFK_WIDE2        0: ranks > 16 on the all-fp32 MFMA kernel only (default: split-operand first product where it pays)
//   NMFK_MFMA_SSE     0: monitored objective of the ranks > 16 on the packed-VALU objective kernel
//   NMFK_HYB          0 / 1: split-operand MFMA half-step off / on for the ranks >= NMFK_HYB_MINK (default: automatic)
//   NMFK_HYB_PHASES   0 / 1: force the matrix-pipe ranks into one launch group that runs first (default: by sweep size)
//   NMFK_HYB_RES      0: no resident form of the split-operand MFMA half-step (short loop dimension: the loop factor in LDS)
//   NMFK_MERGE        g: the ranks <= 16 share g mixed-rank packed-VALU launch groups (default: by restarts per rank)
//   NMFK_REPLAN       0: no re-planning of the launch geometry as restarts retire ("Retire-aware schedule" in nmfk_mu_sweep);
//                     2: re-plan at every tier whatever the sweep's size (tests)
//   NMFK_CLAMP_ALWAYS 1: the clamp pass of every check block looks at every unit (default: only where a fused finish wrote a value below eps())
//   NMFK_SP_BLK       0: sparse X in the gather form only (also: no sliced-ELL copies are built); 2: blocked form whatever the size
//   NMFK_STREAMS      concurrent rank-group streams (8; sparse X: 1);  NMFK_HOST_TIMING=1 prints the host's share of the loop
//   NMFK_COHORTS      c: the matrix-pipe launch group runs as c cohorts of units on c streams (default: by the group's size)
//   NMFK_FUSE_RED     1 (EXPERIMENTAL, off by default): an H half-step whose loop range is split over workgroups gets no reduce launch when the
//                     W half-step behind it runs the resident form -- that launch sums the partial numerators while it stages H.  The new H has
//                     reduce_kernel's bits; rowsum(H) is added in another order (rounding-level differences in the next W half-step's
//                     denominators).  Measured 2.5 % slower on a 60-unit share (profiles/r05/dense_probes.txt), covered by an opt-in test only.
struct Tuning {
  int target_wgs = -1, wide = 1, hyb = -1, hyb_mink = -1, merge = -1, phases = -1;
  int wide_sse = 1, streams = -1, host_timing = 0;
  int hyb_res = 1, wide2 = 1, sp_blk = 1, replan = 1, clamp_always = 0, defer_obj = 1, wide_groups = 2, cohorts = -1, legacy_geo = 0, fuse_red = 0, hyb_lag = -1, wide_bn = 1, debug = 0;
  int exp_geo[2][3] = {{-1, -1, -1}, {-1, -1, -1}};  // NMFK_EXP_GEO="hws,hS,hres,wws,wS,wres": forced geometry of the matrix-pipe group (experiments)
  // (not knobs any more -- round 3's A/B switches with their measured settings: one mixed-rank matrix-pipe group, the waves of
  //  a workgroup may split a loop range 8 ways, the small ranks on their own kernel variants, merged sweeps side by side; the
  //  resident form hands a wave up to eight pairs of lane tiles -- the planner's cost model picks the count: two workgroups per
  //  unit at 480 units of the bench shape, 1.3975 -> 1.3775 ms per iteration, four at 240, profiles/r05/geometry_scan.txt)
  static constexpr int hyb_groups = 1, max_wsplit = 8, hyb_sse = 1, merge_phased = 0, hyb_small = 1, hyb_res_tpw = 8;
};
Tuning read_tuning() {
  Tuning t;
  auto geti = [](const char *name, int &dst) {
    if (const char *e = getenv(name)) dst = atoi(e);
  };
  geti("NMFK_TARGET_WGS", t.target_wgs);
  geti("NMFK_MFMA_WIDE", t.wide);
  geti("NMFK_HYB", t.hyb);
  if (t.hyb > 1) t.hyb = 1;
  geti("NMFK_HYB_MINK", t.hyb_mink);
  geti("NMFK_MERGE", t.merge);
  geti("NMFK_HYB_PHASES", t.phases);
  geti("NMFK_MFMA_SSE", t.wide_sse);
  geti("NMFK_STREAMS", t.streams);
  if (t.streams >= 0) t.streams = std::max(1, std::min(64, t.streams));
  geti("NMFK_HOST_TIMING", t.host_timing);
  geti("NMFK_WIDE2", t.wide2);
  geti("NMFK_SP_BLK", t.sp_blk);
  geti("NMFK_HYB_RES", t.hyb_res);
  geti("NMFK_REPLAN", t.replan);
  geti("NMFK_CLAMP_ALWAYS", t.clamp_always);
  geti("NMFK_DEFER_OBJ", t.defer_obj);
  geti("NMFK_WIDE_GROUPS", t.wide_groups);
  geti("NMFK_COHORTS", t.cohorts);
  if (const char *e = getenv("NMFK_EXP_GEO")) sscanf(e, "%d,%d,%d,%d,%d,%d", &t.exp_geo[0][0], &t.exp_geo[0][1], &t.exp_geo[0][2], &t.exp_geo[1][0], &t.exp_geo[1][1], &t.exp_geo[1][2]);
  geti("NMFK_EXP_LEGACY_GEO", t.legacy_geo);
  geti("NMFK_FUSE_RED", t.fuse_red);
  geti("NMFK_HYB_LAG", t.hyb_lag);
  geti("NMFK_WIDE_BN", t.wide_bn);
  geti("NMFK_DEBUG", t.debug);  // (read once per sweep, here: the planner and the sweep print what they decided; 2: every candidate geometry)
  if (t.cohorts >= 0) t.cohorts = std::max(1, std::min(8, t.cohorts));
  return t;
}

}  // namespace

NMFK_EXPORT int nmfk_version(void) { return 210; }

NMFK_EXPORT const char *nmfk_last_error(void) { return nmfk_error_slot().c_str(); }

NMFK_EXPORT int nmfk_device_count(int *count) {
  if (!count) return fail(NMFK_ERR_BAD_ARG, "count is null");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *count = c;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_mu_default_params(nmfk_mu_params *p) {
  if (!p) return fail(NMFK_ERR_BAD_ARG, "params is null");
  memset(p, 0, sizeof(*p));
  p->tol = 1e-19;
  p->tolOF = 1e-3;
  p->lambda = 1e-32;
  p->weight = 1.0;
  p->maxiter = 10000;
  p->maxreattempts = 2;
  p->maxbaditers = 10;
  p->stopconv = 1000;
  p->Wfixed = 0;
  p->Hfixed = 0;
  p->normalize = 1;
  p->compute = NMFK_COMPUTE_F32;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_create(int device, nmfk_ctx **out) {
  if (!out) return fail(NMFK_ERR_BAD_ARG, "out is null");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(NMFK_ERR_NO_DEVICE, "no HIP device is visible; libnmfk_hip has no CPU fallback");
  if (device < 0 || device >= count) return fail(NMFK_ERR_BAD_ARG, "device index out of range");
  HIPCHECK(hipSetDevice(device));
  nmfk_ctx *ctx = new nmfk_ctx();
  ctx->device = device;
  if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) {
    delete ctx;
    return fail(NMFK_ERR_HIP, "hipGetDeviceProperties failed");
  }
  if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    std::string msg = std::string("device is ") + ctx->prop.gcnArchName + "; libnmfk_hip is built for gfx950 only";
    delete ctx;
    return fail(NMFK_ERR_NO_DEVICE, msg);
  }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return fail(NMFK_ERR_HIP, "hipStreamCreate failed");
  }
  *out = ctx;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_destroy(nmfk_ctx *ctx) {
  if (!ctx) return NMFK_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
  for (hipStream_t gs : ctx->gstreams) (void)hipStreamDestroy(gs);
  if (ctx->poll_stream) (void)hipStreamDestroy(ctx->poll_stream);
  if (ctx->Xc) (void)hipFree(ctx->Xc);
  if (ctx->Xr) (void)hipFree(ctx->Xr);
  if (ctx->Wgt) (void)hipFree(ctx->Wgt);
  free_sparse(ctx);
  ctx->arena.release();
  ctx->scratch.release();
  ctx->xtile.release();
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_device_info(nmfk_ctx *ctx, char *name, int name_len, int *compute_units, int64_t *hbm_bytes) {
  if (!ctx) return fail(NMFK_ERR_BAD_ARG, "ctx is null");
  if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_set_X(nmfk_ctx *ctx, const float *X, int64_t n, int64_t m, int64_t ldx, double lambda,
                           int64_t *nan_count, int64_t *zero_count) {
  if (!ctx || !X) return fail(NMFK_ERR_BAD_ARG, "ctx or X is null");
  if (n <= 0 || m <= 0) return fail(NMFK_ERR_BAD_ARG, "Input array has a zero dimension!");  // Exec:242-244
  if (ldx < n) return fail(NMFK_ERR_BAD_ARG, "ldx < n");
  // the half-step kernels address X with 32-bit byte offsets inside a group of <= 4 rows (buffer loads)
  if (n > (1 << 27) || m > (1 << 27)) return fail(NMFK_ERR_UNSUPPORTED, "dimension exceeds 2^27");
  HIPCHECK(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)n * (size_t)m * sizeof(float);
  if (ctx->Xc) (void)hipFree(ctx->Xc);
  if (ctx->Xr) (void)hipFree(ctx->Xr);
  ctx->Xc = ctx->Xr = nullptr;
  ctx->n = ctx->m = 0;
  if (ctx->Wgt) (void)hipFree(ctx->Wgt);
  ctx->Wgt = nullptr;
  free_sparse(ctx);
  // + slack: the half-step kernels read pairs of adjacent entries; the pair of the last entry pokes 4 bytes past the end
  HIPCHECK(hipMalloc((void **)&ctx->Xc, bytes + 64));
  HIPCHECK(hipMalloc((void **)&ctx->Xr, bytes + 64));
  const size_t inbytes = (size_t)ldx * (size_t)m * sizeof(float);
  if (ctx->scratch.ensure(inbytes + 256)) return fail(NMFK_ERR_HIP, "out of device memory (X staging)");
  unsigned long long *counts = (unsigned long long *)ctx->scratch.p;
  float *Xin = (float *)(ctx->scratch.p + 256);
  HIPCHECK(hipMemsetAsync(counts, 0, 3 * sizeof(unsigned long long), ctx->stream));
  HIPCHECK(hipMemcpyAsync(Xin, X, ((size_t)ldx * (size_t)(m - 1) + (size_t)n) * sizeof(float), hipMemcpyDefault,
                          ctx->stream));
  nmfk_launch_preprocess(Xin, ldx, n, m, (float)lambda, ctx->Xc, ctx->Xr, counts, ctx->stream);
  ctx->xgen++;
  HIPCHECK(hipGetLastError());
  unsigned long long h[3] = {0, 0, 0};
  HIPCHECK(hipMemcpyAsync(h, counts, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  if (h[0] > 0) {
    (void)hipFree(ctx->Xc);
    (void)hipFree(ctx->Xr);
    ctx->Xc = ctx->Xr = nullptr;
    return fail(NMFK_ERR_NEGATIVE, "All matrix entries must be nonnegative!");  // Mult:4-7
  }
  ctx->n = n;
  ctx->m = m;
  ctx->lambda = lambda;
  ctx->nan_count = (int64_t)h[1];
  ctx->zero_count = (int64_t)h[2];
  if (nan_count) *nan_count = ctx->nan_count;
  if (zero_count) *zero_count = ctx->zero_count;
  return NMFK_OK;
}

namespace {
// Sliced ELL of one orientation (NmfkSparseArgs::ell): L lane elements (ptr / idx / val: their records, sorted by index),
// D rows of the gathered factor.  Skipped (the gather form serves) when the slots exceed NMFK_ELL_MAX_PAD x the records:
// a wave walks the longest lane element of its slice, so skewed lane elements would waste its time as well as the memory.
constexpr double NMFK_ELL_MAX_PAD = 4.0;
int build_ell(nmfk_ctx *ctx, int o, int64_t L, int64_t D, const std::vector<int32_t> &ptr, const std::vector<int32_t> &idx,
              const std::vector<float> &val) {
  const int64_t nsl = (L + 63) / 64, ngb = (D + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS, nz = ptr[L];
  if (nz == 0) return NMFK_OK;
  std::vector<int32_t> ep((size_t)(nsl * ngb + 1), 0);
  std::vector<int32_t> cnt((size_t)ngb);
  int64_t rows = 0;
  for (int64_t sl = 0; sl < nsl; ++sl) {
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int64_t l = sl * 64; l < std::min(L, sl * 64 + 64); ++l) {
      int32_t p = ptr[l];
      while (p < ptr[l + 1]) {
        const int64_t b = idx[p] / NMFK_SPB_ROWS;
        int32_t q = p;
        while (q < ptr[l + 1] && idx[q] / NMFK_SPB_ROWS == b) ++q;
        cnt[b] = std::max(cnt[b], q - p);
        p = q;
      }
    }
    for (int64_t b = 0; b < ngb; ++b) {
      ep[(size_t)(sl * ngb + b)] = (int32_t)rows;
      rows += cnt[b];
      if (rows * 64 > (int64_t)(NMFK_ELL_MAX_PAD * nz) + 64 * 1024 || rows > 0x7ffffff0 / 64) return NMFK_OK;  // (too skewed)
    }
  }
  ep[(size_t)(nsl * ngb)] = (int32_t)rows;
  const int64_t pad = 8;  // slot rows the kernel's run-ahead loads may read past the last run
  std::vector<int2> e((size_t)((rows + pad) * 64), int2{-1, 0});
  // Placement of a lane's records inside a run (round 6).  The kernel gathers, per slot row, one staged row per lane with ds_read_b128, which the LDS
  // serves 16 lanes at a time (groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md), each lane on a window of four
  // banks = (row * stride / 4 + chunk) mod 16 with an odd stride / 4 (nmfk_spb_stride): two lanes of a group whose rows agree mod 16 cost a cycle more.
  // Taken in index order the rows of a slot row are a random draw (~1.75 cycles per group and read: half of the LDS's busy cycles were conflicts,
  // profiles/r06/sparse_analysis.txt).  The ORDER of a lane's records inside its run is free, and so is the slot row in which a lane with fewer records
  // than the run sits out: per slot row and group, lanes that must place (as many records left as rows) go first, every lane takes the record of its
  // list whose window is least used so far, the others place only on an unused window.  (Simulated on Poisson(5) runs: 7.0 -> 3.7 LDS cycles per read.)
  static const int kGroup[64] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3};
  std::vector<int32_t> lo(64), hi(64);      // records of lane j in the granule at hand: [lo, hi) of idx / val
  std::vector<int32_t> order(64);
  std::vector<char> used;                   // per record of the granule: placed
  for (int64_t sl = 0; sl < nsl; ++sl) {
    const int nl = (int)(std::min(L, sl * 64 + 64) - sl * 64);
    std::vector<int32_t> pos(nl);
    for (int j = 0; j < nl; ++j) pos[j] = ptr[sl * 64 + j];
    for (int64_t b = 0; b < ngb; ++b) {
      const int64_t r0 = ep[(size_t)(sl * ngb + b)], len = ep[(size_t)(sl * ngb + b + 1)] - r0;
      int32_t base = INT32_MAX, top = 0;
      for (int j = 0; j < nl; ++j) {
        const int64_t l = sl * 64 + j;
        lo[j] = pos[j];
        int32_t q = pos[j];
        while (q < ptr[l + 1] && idx[q] / NMFK_SPB_ROWS == b) ++q;
        hi[j] = pos[j] = q;
        if (hi[j] > lo[j]) base = std::min(base, lo[j]), top = std::max(top, hi[j]);
      }
      if (len <= 0) continue;
      used.assign((size_t)std::max(0, top - base), 0);
      std::vector<int32_t> left(nl);
      for (int j = 0; j < nl; ++j) left[j] = hi[j] - lo[j];
      for (int64_t t = 0; t < len; ++t) {
        const int32_t rows_left = (int32_t)(len - t);
        for (int grp = 0; grp < 4; ++grp) {
          int use[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          int no = 0;
          for (int j = 0; j < nl; ++j)
            if (kGroup[j] == grp && left[j] > 0) order[no++] = j;
          std::stable_sort(order.begin(), order.begin() + no, [&](int a, int c) { return left[a] > left[c]; });
          for (int oi = 0; oi < no; ++oi) {
            const int j = order[oi];
            int32_t best = -1;
            for (int32_t q = lo[j]; q < hi[j]; ++q)
              if (!used[(size_t)(q - base)] && (best < 0 || use[idx[q] & 15] < use[idx[best] & 15])) best = q;
            if (left[j] < rows_left && use[idx[best] & 15] > 0) continue;  // (may wait for a row in which its window is free)
            used[(size_t)(best - base)] = 1;
            ++use[idx[best] & 15];
            --left[j];
            int2 &w = e[(size_t)((r0 + t) * 64 + j)];
            w.x = idx[best];
            memcpy(&w.y, &val[best], 4);
          }
        }
      }
    }
  }
  // no memory for the blocked form is not an error: the gather kernels need none of this
  if (hipMalloc((void **)&ctx->ell[o], sizeof(int2) * e.size()) != hipSuccess ||
      hipMalloc((void **)&ctx->ellptr[o], sizeof(int32_t) * ep.size()) != hipSuccess ||
      hipMemcpy(ctx->ell[o], e.data(), sizeof(int2) * e.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ctx->ellptr[o], ep.data(), sizeof(int32_t) * ep.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (ctx->ell[o]) (void)hipFree(ctx->ell[o]);
    if (ctx->ellptr[o]) (void)hipFree(ctx->ellptr[o]);
    ctx->ell[o] = nullptr;
    ctx->ellptr[o] = nullptr;
    return NMFK_OK;
  }
  ctx->ell_ngb[o] = (int)ngb;
  ctx->ell_pad[o] = (double)(rows * 64) / (double)nz;
  return NMFK_OK;
}
}  // namespace

static int set_X_csc_impl(nmfk_ctx *ctx, int64_t n, int64_t m, int64_t nnz, const int64_t *colptr, const int32_t *rowidx,
                          const float *vals, int64_t *kept);
NMFK_EXPORT int nmfk_set_X_csc(nmfk_ctx *ctx, int64_t n, int64_t m, int64_t nnz, const int64_t *colptr,
                               const int32_t *rowidx, const float *vals, int64_t *kept) {
  try {
    return set_X_csc_impl(ctx, n, m, nnz, colptr, rowidx, vals, kept);
  } catch (const std::bad_alloc &) {  // (host-side CSR twin: no C++ exception crosses the boundary)
    if (ctx) free_sparse(ctx);
    return fail(NMFK_ERR_HIP, "out of host memory while building the CSR twin of X");
  }
}
static int set_X_csc_impl(nmfk_ctx *ctx, int64_t n, int64_t m, int64_t nnz, const int64_t *colptr, const int32_t *rowidx,
                          const float *vals, int64_t *kept) {
  if (!ctx || !colptr || (nnz > 0 && (!rowidx || !vals))) return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (n <= 0 || m <= 0) return fail(NMFK_ERR_BAD_ARG, "Input array has a zero dimension!");
  if (n > 0x7fffff00 || m > 0x7fffff00 || nnz > 0x7fffff00) return fail(NMFK_ERR_UNSUPPORTED, "size exceeds int32 range");
  // (the sparse kernels index a factor's elements with 32 bits: rows * NMFK_MAX_K < 2^31)
  if (n > (1 << 24) || m > (1 << 24)) return fail(NMFK_ERR_UNSUPPORTED, "sparse X: dimension exceeds 2^24");
  if (colptr[0] != 0 || colptr[m] != nnz) return fail(NMFK_ERR_BAD_ARG, "bad colptr");
  HIPCHECK(hipSetDevice(ctx->device));
  // host-side: validate, drop entries <= 0 (they are zeros: Mult:17-18 turns them into lambda), build CSR
  std::vector<int32_t> cp(m + 1, 0), ri, rp(n + 1, 0), ci;
  std::vector<float> vc, vr;
  ri.reserve(nnz);
  vc.reserve(nnz);
  for (int64_t j = 0; j < m; ++j) {
    if (colptr[j + 1] < colptr[j]) return fail(NMFK_ERR_BAD_ARG, "colptr is not monotone");
    for (int64_t p = colptr[j]; p < colptr[j + 1]; ++p) {
      const float v = vals[p];
      const int32_t i = rowidx[p];
      if (i < 0 || i >= n) return fail(NMFK_ERR_BAD_ARG, "row index out of range");
      if (v < 0) return fail(NMFK_ERR_NEGATIVE, "All matrix entries must be nonnegative!");
      if (v != v) return fail(NMFK_ERR_UNSUPPORTED, "NaN (missing) entries need the dense path (nmfk_set_X)");
      if (v > 0) {
        ri.push_back(i);
        vc.push_back(v);
        rp[i + 1]++;
      }
    }
    cp[j + 1] = (int32_t)ri.size();
    // row indices ascending inside a column (a Julia SparseMatrixCSC and scipy's canonical form are; a direct caller of
    // the C ABI need not be): the sliced ELL below walks a lane element's granules in index order, and the CSR twin
    // inherits the order.  Duplicates stay separate records (their contributions add, as in a COO sum).
    const int32_t c0 = cp[j], c1 = cp[j + 1];
    bool sorted = true;
    for (int32_t p = c0 + 1; p < c1 && sorted; ++p) sorted = ri[p - 1] <= ri[p];
    if (!sorted) {
      std::vector<std::pair<int32_t, float>> col((size_t)(c1 - c0));
      for (int32_t p = c0; p < c1; ++p) col[(size_t)(p - c0)] = {ri[p], vc[p]};
      std::stable_sort(col.begin(), col.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
      for (int32_t p = c0; p < c1; ++p) {
        ri[p] = col[(size_t)(p - c0)].first;
        vc[p] = col[(size_t)(p - c0)].second;
      }
    }
  }
  const int64_t nz = (int64_t)ri.size();
  for (int64_t i = 0; i < n; ++i) rp[i + 1] += rp[i];
  ci.resize(nz);
  vr.resize(nz);
  {
    std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
    for (int64_t j = 0; j < m; ++j)
      for (int32_t p = cp[j]; p < cp[j + 1]; ++p) {
        const int32_t q = fill[ri[p]]++;
        ci[q] = (int32_t)j;
        vr[q] = vc[p];
      }
  }
  if (ctx->Xc) (void)hipFree(ctx->Xc);
  if (ctx->Xr) (void)hipFree(ctx->Xr);
  ctx->Xc = ctx->Xr = nullptr;
  if (ctx->Wgt) (void)hipFree(ctx->Wgt);
  ctx->Wgt = nullptr;
  free_sparse(ctx);
  const size_t nzs = (size_t)std::max<int64_t>(nz, 1);
  HIPCHECK(hipMalloc((void **)&ctx->colptr, sizeof(int32_t) * (m + 1)));
  HIPCHECK(hipMalloc((void **)&ctx->rowptr, sizeof(int32_t) * (n + 1)));
  HIPCHECK(hipMalloc((void **)&ctx->rec_csc, sizeof(int2) * nzs));
  HIPCHECK(hipMalloc((void **)&ctx->rec_csr, sizeof(int2) * nzs));
  HIPCHECK(hipMemset(ctx->rec_csc, 0, sizeof(int2) * nzs));
  HIPCHECK(hipMemset(ctx->rec_csr, 0, sizeof(int2) * nzs));
  HIPCHECK(hipMemcpy(ctx->colptr, cp.data(), sizeof(int32_t) * (m + 1), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(ctx->rowptr, rp.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  if (nz > 0) {
    std::vector<int2> rec((size_t)nz);
    for (int64_t p = 0; p < nz; ++p) {
      rec[p].x = ri[p];
      memcpy(&rec[p].y, &vc[p], 4);
    }
    HIPCHECK(hipMemcpy(ctx->rec_csc, rec.data(), sizeof(int2) * nz, hipMemcpyHostToDevice));
    for (int64_t p = 0; p < nz; ++p) {
      rec[p].x = ci[p];
      memcpy(&rec[p].y, &vr[p], 4);
    }
    HIPCHECK(hipMemcpy(ctx->rec_csr, rec.data(), sizeof(int2) * nz, hipMemcpyHostToDevice));
  }
  // blocked form (sp_blk_kernel): the sliced ELL of the rows (W half-step) and of the columns (H half-step)
  if (read_tuning().sp_blk) {  // (NMFK_SP_BLK=0: gather kernels only, no ELL copies)
    try {
      (void)build_ell(ctx, 0, n, m, rp, ci, vr);
      (void)build_ell(ctx, 1, m, n, cp, ri, vc);
    } catch (const std::bad_alloc &) {  // host memory for the ELL build: the gather form serves
    }
  }
  ctx->sparse = true;
  ctx->nnz = nz;
  ctx->n = n;
  ctx->m = m;
  ctx->nan_count = 0;
  ctx->zero_count = n * m - nz;
  if (kept) *kept = nz;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_set_weight(nmfk_ctx *ctx, const float *weight, int64_t n, int64_t m) {
  if (!ctx) return fail(NMFK_ERR_BAD_ARG, "ctx is null");
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->Wgt) (void)hipFree(ctx->Wgt);
  ctx->Wgt = nullptr;
  if (!weight) return NMFK_OK;
  if (!ctx->Xc) return fail(NMFK_ERR_NO_X, "nmfk_set_X has not been called (array weights need the dense path)");
  if (n != ctx->n || m != ctx->m) return fail(NMFK_ERR_BAD_ARG, "weight must have the size of X");
  HIPCHECK(hipMalloc((void **)&ctx->Wgt, sizeof(float) * (size_t)n * (size_t)m));
  HIPCHECK(hipMemcpyAsync(ctx->Wgt, weight, sizeof(float) * (size_t)n * (size_t)m, hipMemcpyDefault, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_fill_uniform(nmfk_ctx *ctx, uint64_t seed, uint64_t offset, int64_t count, float *out) {
  if (!ctx || !out || count < 0) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  if (count == 0) return NMFK_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->scratch.ensure((size_t)count * sizeof(float))) return fail(NMFK_ERR_HIP, "out of device memory");
  nmfk_launch_fill_uniform(seed, offset, count, (float *)ctx->scratch.p, ctx->stream);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(out, ctx->scratch.p, (size_t)count * sizeof(float), hipMemcpyDefault, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_robustkmeans(nmfk_ctx *ctx, int d, int64_t n64, const float *X, int k, int repeats, int maxiter, double tol,
                                  uint64_t seed, int32_t *assignments, float *centers, float *costs, int32_t *counts,
                                  double *totalcost, int32_t *best_repeat, int32_t *iterations, int32_t *nclusters,
                                  double *all_costs, float *silhouettes) {  // (the signature of ABI 200; callers built against it keep working)
  return nmfk_robustkmeans_ex(ctx, d, n64, X, k, repeats, maxiter, tol, seed, assignments, centers, costs, counts, totalcost,
                              best_repeat, iterations, nclusters, all_costs, silhouettes, nullptr);
}
NMFK_EXPORT int nmfk_robustkmeans_ex(nmfk_ctx *ctx, int d, int64_t n64, const float *X, int k, int repeats, int maxiter,
                                  double tol, uint64_t seed, int32_t *assignments, float *centers, float *costs,
                                  int32_t *counts, double *totalcost, int32_t *best_repeat, int32_t *iterations,
                                  int32_t *nclusters, double *all_costs, float *silhouettes, int32_t *converged) {
  if (!ctx) return NMFK_ERR_BAD_ARG;
  if (!X || !assignments || !centers || !costs || !counts || !totalcost) return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (d <= 0 || n64 <= 0 || n64 > (1 << 27) || repeats <= 0 || maxiter < 0) return fail(NMFK_ERR_BAD_ARG, "bad dimensions");
  const int n = (int)n64;
  if (k < 1 || k > n) return fail(NMFK_ERR_BAD_ARG, "k must be from 1:n");  // Clustering.kmeans argument check
  if (k > NMFK_MAX_K || (size_t)d * k > 8192) return fail(NMFK_ERR_BAD_ARG, "k <= 64 and d*k <= 8192 supported");
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  Bump B;
  const size_t oX = B.take(sizeof(float) * (size_t)d * n), oZ = B.take(sizeof(float) * (size_t)d * n);
  const size_t oA = B.take(sizeof(int32_t) * (size_t)repeats * n), oC = B.take(sizeof(float) * (size_t)repeats * n);
  const size_t oW = B.take(sizeof(float) * (size_t)repeats * n), oCe = B.take(sizeof(float) * (size_t)repeats * d * k);
  const size_t oCn = B.take(sizeof(int32_t) * (size_t)repeats * k), oT = B.take(sizeof(double) * repeats);
  const size_t oI = B.take(sizeof(int32_t) * repeats), oV = B.take(sizeof(int32_t) * repeats);
  const size_t oSa = B.take(sizeof(int32_t) * (size_t)n), oSc = B.take(sizeof(int32_t) * k), oS = B.take(sizeof(float) * (size_t)n);
  if (ctx->scratch.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory (k-means workspace)");
  char *S = ctx->scratch.p;
  HIPCHECK(hipMemcpyAsync(S + oX, X, sizeof(float) * (size_t)d * n, hipMemcpyDefault, st));
  nmfk_launch_kmeans((const float *)(S + oX), d, n, k, repeats, maxiter, tol, seed, (int32_t *)(S + oA), (float *)(S + oC),
                     (float *)(S + oW), (float *)(S + oCe), (int32_t *)(S + oCn), (double *)(S + oT), (int32_t *)(S + oI),
                     (int32_t *)(S + oV), st);
  HIPCHECK(hipGetLastError());
  std::vector<double> tot(repeats);
  std::vector<int32_t> its(repeats), cvg(repeats);
  HIPCHECK(hipMemcpyAsync(tot.data(), S + oT, sizeof(double) * repeats, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipMemcpyAsync(its.data(), S + oI, sizeof(int32_t) * repeats, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipMemcpyAsync(cvg.data(), S + oV, sizeof(int32_t) * repeats, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipStreamSynchronize(st));
  int best = 0;
  for (int r = 1; r < repeats; ++r)
    if (tot[r] < tot[best]) best = r;  // Clus:227: strict <, the first of equal costs wins
  if (all_costs) memcpy(all_costs, tot.data(), sizeof(double) * repeats);
  std::vector<int32_t> a(n), cn(k);
  std::vector<float> ce((size_t)d * k);
  HIPCHECK(hipMemcpyAsync(a.data(), S + oA + sizeof(int32_t) * (size_t)best * n, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipMemcpyAsync(costs, S + oC + sizeof(float) * (size_t)best * n, sizeof(float) * (size_t)n, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(ce.data(), S + oCe + sizeof(float) * (size_t)best * d * k, sizeof(float) * (size_t)d * k, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipMemcpyAsync(cn.data(), S + oCn + sizeof(int32_t) * (size_t)best * k, sizeof(int32_t) * k, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipStreamSynchronize(st));
  // sortclustering (Clus:264-292): clusters in order of first appearance, then stably by decreasing count
  std::vector<int> first, newlab(k, 0);
  {
    std::vector<char> seen(k, 0);
    for (int j = 0; j < n; ++j)
      if (!seen[a[j]]) {
        seen[a[j]] = 1;
        first.push_back(a[j]);
      }
    std::stable_sort(first.begin(), first.end(), [&](int x, int y) { return cn[x] > cn[y]; });
    for (size_t q = 0; q < first.size(); ++q) newlab[first[q]] = (int)q + 1;
  }
  const int kf = (int)first.size();
  for (int j = 0; j < n; ++j) assignments[j] = newlab[a[j]];
  for (int c = 0; c < k; ++c) {
    counts[c] = c < kf ? cn[first[c]] : 0;
    for (int i = 0; i < d; ++i) centers[i + (size_t)c * d] = c < kf ? ce[i + (size_t)first[c] * d] : 0.f;
  }
  *totalcost = tot[best];
  if (best_repeat) *best_repeat = best;
  if (iterations) *iterations = its[best];
  if (converged) *converged = cvg[best] != 0;
  if (nclusters) *nclusters = kf;
  if (silhouettes) {
    int amax = 0;
    for (int j = 0; j < n; ++j) amax = std::max(amax, (int)assignments[j]);
    if (amax > 1) {  // Clus:211-218
      HIPCHECK(hipMemcpyAsync(S + oSa, assignments, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
      HIPCHECK(hipMemcpyAsync(S + oSc, counts, sizeof(int32_t) * k, hipMemcpyHostToDevice, st));
      nmfk_launch_point_silhouettes((const float *)(S + oX), d, n, (const int32_t *)(S + oSa), (const int32_t *)(S + oSc), k,
                                    (float *)(S + oZ), (float *)(S + oS), st);
      HIPCHECK(hipGetLastError());
      HIPCHECK(hipMemcpyAsync(silhouettes, S + oS, sizeof(float) * (size_t)n, hipMemcpyDefault, st));
      HIPCHECK(hipStreamSynchronize(st));
    } else {
      for (int j = 0; j < n; ++j) silhouettes[j] = 0.f;
    }
  }
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_set_profiling(nmfk_ctx *ctx, int enabled) {
  if (!ctx) return fail(NMFK_ERR_BAD_ARG, "ctx is null");
  ctx->profiling = enabled != 0;
  ctx->prof.clear();
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_set_objective_trace(nmfk_ctx *ctx, int enabled) {
  if (!ctx) return fail(NMFK_ERR_BAD_ARG, "ctx is null");
  ctx->trace_objective = enabled != 0;
  ctx->obj_trace.clear();
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_get_objective_trace(nmfk_ctx *ctx, int kidx, int restart, double *out, int cap, int *count) {
  if (!ctx || !count || cap < 0 || (cap > 0 && !out)) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  *count = 0;
  if (ctx->obj_trace.empty()) return fail(NMFK_ERR_BAD_ARG, "no objective trace: nmfk_set_objective_trace before the sweep");
  const size_t idx = (size_t)kidx * ctx->obj_trace_nruns + restart;
  if (kidx < 0 || restart < 0 || restart >= ctx->obj_trace_nruns || idx >= ctx->obj_trace_unit.size() || ctx->obj_trace_unit[idx] < 0)
    return fail(NMFK_ERR_BAD_ARG, "no such unit in the last sweep");
  const double *t = ctx->obj_trace.data() + (size_t)ctx->obj_trace_unit[idx] * ctx->obj_trace_stride;
  int n = 0;
  while (n < ctx->obj_trace_stride && t[n] == t[n]) ++n;  // (slots of checks that never happened hold NaN bits)
  for (int i = 0; i < n && i < cap; ++i) out[i] = t[i];
  *count = n;
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_last_sweep_info(nmfk_ctx *ctx, int32_t info[8]) {
  if (!ctx || !info) return fail(NMFK_ERR_BAD_ARG, "null argument");
  memcpy(info, ctx->sweep_info, 8 * sizeof(int32_t));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_last_sweep_info_ex(nmfk_ctx *ctx, int32_t *info, int count) {
  if (!ctx || !info || count < 0 || count > 16) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  memcpy(info, ctx->sweep_info, (size_t)count * sizeof(int32_t));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_get_profile(nmfk_ctx *ctx, int max_entries, char (*names)[64], double *total_ms, int64_t *launches,
                                 double *flops, int *count) {
  if (!ctx || !count) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  int c = 0;
  for (const auto &kv : ctx->prof) {
    if (c >= max_entries) break;
    if (names) snprintf(names[c], 64, "%s", kv.first.c_str());
    if (total_ms) total_ms[c] = kv.second.ms;
    if (launches) launches[c] = kv.second.launches;
    if (flops) flops[c] = kv.second.flops;
    ++c;
  }
  *count = c;
  return NMFK_OK;
}

// --------------------------------------------------------------------------------------------------------
// the sweep
// --------------------------------------------------------------------------------------------------------
namespace {
// Launch geometry of a launch group of `units` units on the matrix-pipe kernels (ranks 2..16, dense fp32) at an n x m matrix on a GPU
// of `cus` CUs: pure host arithmetic, the rule nmfk_mu_sweep applies to such a group (its tier 0) and to every later tier of the
// retire-aware schedule.  [0] = H half-step (lanes = m columns, loop = n rows), [1] = W half-step.
// Workgroups of a matrix-pipe half-step below which the waves of a workgroup take loop ranges of their own (wsplit = 8, every wave
// staging for itself) instead of sharing staged blocks, the loop range split over workgroups (S > 1, reduce_kernel finishes).  Round 3
// put the line at 1.5 per CU (240 units x 2 lane tiles: 0.376 ms shared against 0.436 ms per-wave); measured again with the kernels
// as they are (profiles/r04/few_units_shared_staging.txt, k = 2:16 at 8192 x 512): the shared form wins down to ~90 units (120 units:
// 0.40-0.41 against 0.44-0.46 ms per iteration, 90: 0.32-0.33 against 0.34-0.36), the two are equal at 30-75 and the per-wave form
// wins below (15 units: 0.111 against 0.116, one unit: 0.064 against 0.071): 0.7 per CU.
static int hyb_target_ws(int cus) { return 7 * cus / 10; }

// ---- Launch geometry of a matrix-pipe launch group (ranks 2..16, dense fp32), round 5: a cost model instead of thresholds.
// Rounds 2-4 chose the geometry by rules of thumb found at one or two sizes (workgroups per CU below which the waves of a workgroup
// take loop ranges of their own, "fill the chip four times", "at least two pairs per wave") -- right at 480 and 240 units of the
// bench shape, up to 45 % off below (profiles/r05/geometry_scan.txt: 60 units 0.232 -> 0.191 ms per iteration with the best geometry
// by exhaustive scan, 15 units 0.110 -> 0.074, one unit of k = 8 81 -> 45 us).  What the scan showed is what a GPU launch is: a list of
// workgroups handed, in grid order, to whichever CU slot is free -- so the time of a candidate geometry is the makespan of that list
// schedule, and it can be computed on the host: workgroup time = a fixed part + chunks of 16 loop steps x the time per chunk of the
// unit's kernel variant (k <= 4 / <= 8 / <= 16: 0.39 / 0.57 / 0.75 us per chunk and workgroup on a full CU), units in list order (widest
// ranks first).  The constants are a least-squares fit to 58 measured launches of 30..480 units (tools/planner/fit_model.py: rms error
// 5-6 % of a launch's duration; profiles/r05/planner_fit.txt).  Candidates: the streaming form with shared staging and the loop range split
// S ways over workgroups (S > 1: + reduce_kernel), the same with per-wave loop ranges (wsplit), and -- where the loop factor fits the
// LDS -- the resident form with g workgroups per unit.  The cheapest wins; ties go to fewer workgroups.
struct HybMix {
  int n16 = 0, n8 = 0, n4 = 0;  // units per kernel variant; list order: 16, 8, 4 (ranks descending)
  int units() const { return n16 + n8 + n4; }
  int variant_of(int u) const { return u < n16 ? 16 : u < n16 + n8 ? 8 : 4; }
  int vmax() const { return n16 ? 16 : n8 ? 8 : 4; }
  HybMix scaled(int c) const {  // the same mix for c units (later tiers of the retire-aware schedule); the widest variant stays in:
    const int tot = std::max(1, units());  // the launches' LDS is sized for it whatever is left of the list
    HybMix r;
    r.n16 = n16 ? std::max(1, (int)(((int64_t)n16 * c + tot / 2) / tot)) : 0;
    r.n16 = std::min(r.n16, c);
    r.n8 = n8 ? std::min(c - r.n16, std::max((n16 || c - r.n16 == 0) ? 0 : 1, (int)(((int64_t)n8 * c + tot / 2) / tot))) : 0;
    r.n4 = c - r.n16 - r.n8;
    if (n4 == 0 && r.n4 > 0) (n8 ? r.n8 : r.n16) += r.n4, r.n4 = 0;
    return r;
  }
};
static double hyb_chunk_us(int variant) { return 0.75 * (variant <= 4 ? 0.52 : variant <= 8 ? 0.76 : 1.0); }  // per chunk of 16 loop steps and workgroup, full CU
// makespan of a launch: workgroups in grid order (wpu per unit, unit u's take t[u] us) onto `slots` slots; also the busy fraction
static double list_makespan(const std::vector<double> &t, int wpu, int slots, double *busy_frac) {
  const int64_t total = (int64_t)t.size() * wpu;
  double sum = 0, tmax = 0;
  for (double x : t) sum += x * wpu, tmax = std::max(tmax, x);
  double ms;
  if (total <= slots) {
    ms = tmax;
  } else if (total > 60000) {  // (too many to walk: the bound of list scheduling)
    ms = sum / slots + tmax * (1.0 - 1.0 / slots);
  } else {
    std::vector<double> heap((size_t)slots, 0.0);  // min-heap of the slots' free times
    auto cmp = [](double a, double b) { return a > b; };
    for (size_t u = 0; u < t.size(); ++u)
      for (int w = 0; w < wpu; ++w) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        heap.back() += t[u];
        std::push_heap(heap.begin(), heap.end(), cmp);
      }
    ms = *std::max_element(heap.begin(), heap.end());
  }
  if (busy_frac) *busy_frac = ms > 0 ? sum / (ms * slots) : 1.0;
  return ms;
}
// The same for workgroups that SHARE a CU (the streaming kernels: up to `wpc` workgroups of 16 / wpc waves per CU): a workgroup runs
// faster the fewer neighbours it has -- two of eight waves side by side take 0.84 us per chunk each, one alone 0.45 (CU throughput
// 1 : 0.935) -- so the tail of a launch is shorter than fixed slot times make it.  Event simulation per CU: resident workgroups
// progress at rate(residents), a finished one is replaced by the next of the list.  t[u]: us of unit u's workgroups on a FULL CU.
static double cu_share_makespan(const std::vector<double> &t, int wpu, int cus, int wpc, double *busy_frac) {
  const int64_t total = (int64_t)t.size() * wpu;
  double sum = 0, tmax = 0;
  for (double x : t) sum += x * wpu, tmax = std::max(tmax, x);
  auto rate = [&](int residents) { return 1.0 / (0.2 + 0.8 * (double)residents / wpc); };  // per workgroup, 1 = full CU
  double ms;
  if (total > 60000) {
    ms = sum / ((double)cus * wpc) + tmax * (1.0 - 1.0 / ((double)cus * wpc));
  } else {
    struct Cu {
      double rem[4];
      int nres;
      double last;
    };
    std::vector<Cu> cu((size_t)cus, Cu{{0, 0, 0, 0}, 0, 0.0});
    size_t next_u = 0;
    int next_w = 0;
    auto take = [&](double &work) {
      if (next_u >= t.size()) return false;
      work = t[next_u];
      if (++next_w == wpu) next_w = 0, ++next_u;
      return true;
    };
    // the dispatcher fills the CUs breadth first: one workgroup each, then the second slots, ...
    for (int slot = 0; slot < wpc; ++slot)
      for (int c = 0; c < cus; ++c) {
        double w;
        if (!take(w)) break;
        cu[(size_t)c].rem[cu[(size_t)c].nres++] = w;
      }
    typedef std::pair<double, int> Ev;  // (time of the CU's next completion, CU)
    std::vector<Ev> heap;
    auto cmp = [](const Ev &a, const Ev &b) { return a.first > b.first; };
    auto next_done = [&](const Cu &q) {
      double mn = 1e300;
      for (int i = 0; i < q.nres; ++i) mn = std::min(mn, q.rem[i]);
      return q.last + mn / rate(q.nres);
    };
    for (int c = 0; c < cus; ++c)
      if (cu[(size_t)c].nres) heap.push_back({next_done(cu[(size_t)c]), c});
    std::make_heap(heap.begin(), heap.end(), cmp);
    ms = 0;
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      const Ev e = heap.back();
      heap.pop_back();
      Cu &q = cu[(size_t)e.second];
      const double done = (e.first - q.last) * rate(q.nres);
      int keep = 0;
      for (int i = 0; i < q.nres; ++i) {
        const double r = q.rem[i] - done;
        if (r > 1e-9) q.rem[keep++] = r;
      }
      q.nres = keep;
      q.last = e.first;
      ms = std::max(ms, e.first);
      double w;
      while (q.nres < wpc && take(w)) q.rem[q.nres++] = w;
      if (q.nres) {
        heap.push_back({next_done(q), e.second});
        std::push_heap(heap.begin(), heap.end(), cmp);
      }
    }
  }
  if (busy_frac) *busy_frac = ms > 0 ? sum / (ms * cus * wpc) : 1.0;
  return ms;
}
// streaming form: lanes L, loop D, wsplit ws (1: the eight waves of a workgroup share staged blocks and 256 lanes; > 1: they share 32
// lanes and split the workgroup's loop range), S splits of the loop range over workgroups.  us per launch (+ reduce_kernel)
static double hyb_stream_cost(const HybMix &mix, int L, int D, int cus, int ws, int S, double *busy) {
  const int lt = nmfk_hyb_lane_tile(ws), ntile = (L + lt - 1) / lt;
  int dchunk = (D + S - 1) / S;
  if (S > 1) dchunk = (dchunk + 15) & ~15;
  const int nch = ws > 1 ? ((((dchunk + ws - 1) / ws) + 15) >> 4) : ((dchunk + 15) >> 4);
  const int wpc = ws == 4 ? 4 : 2;  // workgroups per CU (16 waves of 128 registers)
  const double perwave = ws > 1 ? 1.3 : 1.0;  // per-wave staging (60 units, wsplit 8: 125 us)
  std::vector<double> t((size_t)mix.units());
  for (int u = 0; u < mix.units(); ++u) t[(size_t)u] = 6.0 + nch * hyb_chunk_us(mix.variant_of(u)) * perwave;
  double c = 5.0 + cu_share_makespan(t, ntile * S, cus, wpc, busy);
  if (S > 1) c += 5.0 + (double)mix.units() * (S + 2) * L * 10.0 * 4.0 / 3.0e6;  // reduce_kernel: a launch + the partials (9 us at 60 units, S = 8)
  return c;
}
// resident form: g workgroups (16 waves, one per CU) per unit stage the loop factor and walk ceil(pairs / (16 g)) pairs of lane tiles
// per wave (30 units at 8192 x 512, g = 4 / 6 / 8: 114 / 89 / 65 us: 11 us of staging + 24 us per pair of 32 chunks at k = 16)
static double hyb_res_cost(const HybMix &mix, int L, int D, int cus, int g, double *busy) {
  const int ntp = (L + 31) / 32, rw = nmfk_hyb_resident_waves(), nch = ((D + 63) & ~63) >> 4;
  const int pairs = (ntp + rw * g - 1) / (rw * g);
  std::vector<double> t((size_t)mix.units());
  for (int u = 0; u < mix.units(); ++u) {
    const int v = mix.variant_of(u);
    t[(size_t)u] = std::max(6.0, 2.0 + 9.0 * (D / 512.0) * (v / 16.0)) + pairs * (nch * hyb_chunk_us(v) * 0.987 + 0.5);
  }
  return 5.0 + list_makespan(t, g, cus, busy);
}

struct HybPlan {
  int units;
  int res[2];     // workgroups per unit of the resident form (0: streaming form)
  int wsplit[2];  // waves of a workgroup that split the loop range (1: they share staged blocks)
  int S[2];       // splits of the loop range over workgroups (> 1: partial numerators + reduce_kernel)
  int dchunk[2], fused[2];
  int slots[2];   // sum-table slots the half-step's helper kernels cover
  int ns[2];      // sum-table slots a unit's own kernels write
  double us[2], busy[2];  // the model's time of a launch and the fraction of the CU slots it keeps busy
};
// rounds 3-4's resident-form rule (NMFK_EXP_LEGACY_GEO=1, A/B measurements)
static int hyb_res_wgs_legacy(int L, int D, int cus, int vmax, int units) {
  if (units <= 0 || nmfk_hyb_resident_lds(vmax, D) == 0) return 0;
  const int ntp = (L + 31) / 32, rw = nmfk_hyb_resident_waves(), res_tpw = 4;
  const int fill = (4 * cus + units - 1) / units;
  const int gmax = std::max(1, ntp / (2 * rw)), gmin = std::min(gmax, std::max(std::max(1, ntp / (rw * res_tpw)), fill));
  int best = gmin;
  double waste = 1e30;
  for (int gq = gmin; gq <= gmax; ++gq) {
    const int rounds = (ntp + rw * gq - 1) / (rw * gq);
    const double wq = (double)rounds * rw * gq / ntp;
    if (wq < waste - 1e-9) {
      waste = wq;
      best = gq;
    }
  }
  return best;
}
// [0] = H half-step (lanes = m columns, loop = n rows), [1] = W half-step
HybPlan plan_hyb_group(int n, int m, int cus, const HybMix &mix, int target_wgs, bool hyb_res, bool legacy = false,
                       const int (*exp_geo)[3] = nullptr, bool one_round = false, bool dbg = false) {
  HybPlan p;
  const int units = std::max(1, mix.units()), vmax = mix.vmax();
  p.units = units;
  const int target = target_wgs > 0 ? target_wgs : 2 * cus;
  const int target_ws = target_wgs > 0 ? target : hyb_target_ws(cus);
  const int max_ws = Tuning::max_wsplit;
  for (int which = 0; which < 2; ++which) {
    const int L = which == 0 ? m : n, D = which == 0 ? n : m;
    const int wsN = (max_ws >= 8 && D >= 8 * 64) ? 8 : 4;
    p.us[which] = 0, p.busy[which] = 1;
    int res = 0, ws = 1, S = 1;
    if (legacy || target_wgs > 0) {  // (NMFK_TARGET_WGS keeps the threshold rule: tests force split geometries with it)
      res = hyb_res ? hyb_res_wgs_legacy(L, D, cus, vmax, units) : 0;
      auto tiles = [&](int w) { return res > 0 ? res : (L + nmfk_hyb_lane_tile(w) - 1) / nmfk_hyb_lane_tile(w); };
      if ((int64_t)tiles(1) * units < target_ws) ws = wsN;
      const int64_t have = std::max<int64_t>((int64_t)tiles(ws) * units, 1);
      S = (int)((target + have - 1) / have);
    } else {
      double best = 1e30;
      const bool can_res = hyb_res && nmfk_hyb_resident_lds(vmax, D) != 0;
      if (can_res) {
        // As a COHORT's launch (another cohort's launches run beside it) the resident form does best with ONE round of workgroups --
        // measured: cohorts of 60 / 30 / 15 units: 4 / 8 / 16 workgroups per unit, ~240 on 256 CUs each time, where a launch alone
        // on the chip wants ~480 (profiles/r05/geometry_scan.txt) -- the other cohort is the second round.
        const int ntp = (L + 31) / 32, rw = nmfk_hyb_resident_waves(), gmax = std::max(1, ntp / rw);
        const int gmin = std::max(1, ntp / (rw * Tuning::hyb_res_tpw));
        // (candidates: gmin doubled up to gmax.  Workgroups go to the XCDs round-robin, so a count that is a multiple of 8 keeps lane
        //  tile t of EVERY unit on XCD t mod 8 and an XCD's L2 holds an eighth of X; an odd count spreads every unit's tiles over all
        //  eight -- 65536 x 256 (X = 67 MB), 40 units, 19 instead of 32 workgroups per unit: 0.39 -> 0.54 ms per iteration)
        for (int g = gmin; g <= gmax; g = (g < gmax && 2 * g > gmax) ? gmax : 2 * g) {
          if (one_round && g > gmin && (int64_t)g * units > cus) break;
          double b;
          const double c = hyb_res_cost(mix, L, D, cus, g, &b);
          if (dbg) fprintf(stderr, "[nmfk]   %c half-step, %d units: resident g %d: %.1f us (busy %.2f)\n", "HW"[which], units, g, c, b);
          if (c < best * 0.97) best = c, res = g, p.busy[which] = b;
        }
      }
      // (where the loop factor fits the LDS the resident form is the better kernel from a handful of units up; a unit or two are
      //  served faster by the streaming form with the loop range split: more, shorter workgroups)
      if (!can_res || units <= 2) {
        static const int Ss[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64};
        for (int w : {1, wsN})
          for (int Sq : Ss) {
            if (Sq > std::max(1, D / (64 * w))) break;
            // (a launch of more than ~8 rounds of workgroups gains nothing from more, shorter ones: their fixed parts add up --
            //  and the simulation of 60 000 workgroups per candidate would cost the host 0.3 s per sweep)
            const int lt = nmfk_hyb_lane_tile(w);
            if (Sq > 1 && (int64_t)units * ((L + lt - 1) / lt) * Sq > (int64_t)16 * cus) break;
            double b;
            const double c = hyb_stream_cost(mix, L, D, cus, w, Sq, &b);
            if (dbg) fprintf(stderr, "[nmfk]   %c half-step, %d units: streaming wsplit %d S %d: %.1f us (busy %.2f)\n", "HW"[which], units, w, Sq, c, b);
            if (c < best * 0.97) best = c, ws = w, S = Sq, res = 0, p.busy[which] = b;
          }
      }
      p.us[which] = best;
    }
    if (exp_geo && exp_geo[which][2] >= 0 && nmfk_hyb_resident_lds(vmax, D) != 0) res = exp_geo[which][2];
    if (exp_geo && exp_geo[which][0] > 0) ws = exp_geo[which][0];
    if (exp_geo && exp_geo[which][1] > 0) S = exp_geo[which][1];
    p.res[which] = res;
    p.wsplit[which] = ws;
    auto tiles = [&](int w) { return p.res[which] > 0 ? p.res[which] : (L + nmfk_hyb_lane_tile(w) - 1) / nmfk_hyb_lane_tile(w); };
    const int maxS = std::max(1, D / (64 * ws));
    p.S[which] = std::max(1, std::min(S, maxS));
    p.dchunk[which] = (D + p.S[which] - 1) / p.S[which];
    if (p.S[which] > 1) {
      p.dchunk[which] = (p.dchunk[which] + 15) & ~15;
      p.S[which] = (D + p.dchunk[which] - 1) / p.dchunk[which];
    }
    p.fused[which] = p.S[which] == 1;
    p.slots[which] = tiles(ws);
    if (!p.fused[which]) p.slots[which] = std::max(p.slots[which], std::min(64, (L + 31) / 32));
    p.ns[which] = (p.fused[which] || p.res[which] > 0) ? tiles(ws) : p.slots[which];
  }
  return p;
}
// Cohorts of a matrix-pipe launch group (nmfk_mu_sweep, "Cohorts"; NMFK_COHORTS overrides): two when the plan's launches leave CU slots
// idle (tails, rounds of unequal workgroups), which a second stream of launches fills.  One
//  * when a launch fills the chip for many rounds anyway: two kernels side by side then only fragment the CUs (65536 x 256 at 240 units:
//    + 5 %, at 480: + 12 %);
//  * when the launches are short -- below ~80 us per iteration the launch count is what costs (300 x 300: + 4 %, 2048 x 2048,
//    k = 16 x 10: + 14 %, 8192 x 512, k = 16 x 10: 71 -> 80 us);
//  * when a streaming launch has no more workgroups than CUs: each then has a CU to itself, at 1.87x the speed of two side by side,
//    and the workgroups of two such launches from two queues land on the SAME CUs (2048 x 2048, k = 2:16 x 2: + 20 %).
static int nmfk_default_cohorts(const HybPlan &p, int64_t n, int64_t m, int cus) {
  if (p.units < 4 || n * m < 100000 || p.us[0] + p.us[1] < 80.0) return 1;
  const double busy = (p.busy[0] * p.us[0] + p.busy[1] * p.us[1]) / (p.us[0] + p.us[1]);
  for (int w = 0; w < 2; ++w) {
    const int64_t L = w == 0 ? m : n, lt = nmfk_hyb_lane_tile(p.wsplit[w]);
    if (p.res[w] == 0 && (L + lt - 1) / lt * p.S[w] * p.units < 3 * cus / 2) return 1;  // (workgroups of the streaming launch)
  }
  return busy < 0.9 ? 2 : 1;
}
// The plan of a group WITH its cohorts: the group as one launch decides the cohorts; with c > 1 a launch holds units / c of them, so the
// geometry is the plan of such a launch (finer: 60 units as two cohorts run the plan of 30 -- S = 8, eight resident workgroups per
// unit -- which the exhaustive scan found best for them: 0.2317 -> 0.1905 ms per iteration, profiles/r05/geometry_scan.txt).
struct HybCohortPlan {
  HybPlan plan;
  int cohorts;
};
// The planner is a pure function of its arguments and costs ~2 ms of host time per call (list-schedule simulations of every candidate; a sweep with its
// tiers makes ~10 calls: ~20 ms at the bench shape, comparable with the GPU time of a short sweep -- ADVICE r5): the plans are memoised per process.
static HybCohortPlan plan_hyb_cohorts_uncached(int n, int m, int cus, const HybMix &mix, const Tuning &T, bool hyb_res);
static HybCohortPlan plan_hyb_cohorts(int n, int m, int cus, const HybMix &mix, const Tuning &T, bool hyb_res) {
  if (T.debug >= 2) return plan_hyb_cohorts_uncached(n, m, cus, mix, T, hyb_res);  // (the candidates are printed while they are costed)
  static std::mutex mu;
  static std::map<std::array<int, 16>, HybCohortPlan> memo;
  const std::array<int, 16> key = {n, m, cus, mix.n16, mix.n8, mix.n4, T.target_wgs, T.legacy_geo, T.cohorts, hyb_res ? 1 : 0,
                                   T.exp_geo[0][0], T.exp_geo[0][1], T.exp_geo[0][2], T.exp_geo[1][0], T.exp_geo[1][1], T.exp_geo[1][2]};
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
  }
  const HybCohortPlan r = plan_hyb_cohorts_uncached(n, m, cus, mix, T, hyb_res);
  std::lock_guard<std::mutex> lk(mu);
  if (memo.size() > 4096) memo.clear();
  memo[key] = r;
  return r;
}
static HybCohortPlan plan_hyb_cohorts_uncached(int n, int m, int cus, const HybMix &mix, const Tuning &T, bool hyb_res) {
  HybCohortPlan r;
  r.plan = plan_hyb_group(n, m, cus, mix, T.target_wgs, hyb_res, T.legacy_geo != 0, T.exp_geo, false, T.debug >= 2);
  const bool model = !(T.legacy_geo || T.target_wgs > 0);
  r.cohorts = T.cohorts > 0 ? T.cohorts : model ? nmfk_default_cohorts(r.plan, n, m, cus) : 1;
  r.cohorts = std::max(1, std::min(r.cohorts, mix.units()));
  if (r.cohorts > 1 && model) {
    const int units = mix.units();
    r.plan = plan_hyb_group(n, m, cus, mix.scaled((units + r.cohorts - 1) / r.cohorts), T.target_wgs, hyb_res, false, T.exp_geo, true, T.debug >= 2);
    r.plan.units = units;
  }
  return r;
}
}  // namespace

// Test hook (no device needed): the tiers nmfk_mu_sweep plans for a sweep of `units` units whose ranks (all in 2..16, widest kernel
// variant `variant` = 4 / 8 / 16) run in one launch group on the matrix-pipe kernels: tier j for ceil(units / 2^j) units.  Row j of
// `out` (16 ints per row, at most `cap` rows): units, then for the H and the W half-step: res, wsplit, S, dchunk, fused, slots, ns;
// out[15] = cohorts the group would run as with that many units (nmfk_mu_sweep, "Cohorts").  *count = tiers.  The retire-aware schedule switches to tier j when the units still active are <= its `units`.
NMFK_EXPORT int nmfk_plan_hyb_tiers(int64_t n, int64_t m, int variant, int units, int cus, int32_t *out, int cap, int *count) {
  if (!out || !count || cap < 1 || n < 16 || m < 16 || units < 1 || cus < 1) return fail(NMFK_ERR_BAD_ARG, "bad argument");
  const Tuning T = read_tuning();
  int rows = 0;
  for (int c = units;; c = (c + 1) / 2) {
    if (rows >= cap) break;
    HybMix mix;
    if (variant == 0) {  // the ranks 2..16 in equal numbers (the bench sweep): 8 : 4 : 3 of the variants 16 : 8 : 4
      HybMix all;
      all.n16 = 8, all.n8 = 4, all.n4 = 3;
      mix = all.scaled(c);
    } else {
      (variant <= 4 ? mix.n4 : variant <= 8 ? mix.n8 : mix.n16) = c;
    }
    const HybCohortPlan cp = plan_hyb_cohorts((int)n, (int)m, cus, mix, T, T.hyb_res != 0);
    const HybPlan &p = cp.plan;
    int32_t *o = out + 16 * rows++;
    o[0] = p.units;
    for (int w = 0; w < 2; ++w) {
      int32_t *q = o + 1 + 7 * w;
      q[0] = p.res[w], q[1] = p.wsplit[w], q[2] = p.S[w], q[3] = p.dchunk[w], q[4] = p.fused[w], q[5] = p.slots[w], q[6] = p.ns[w];
    }
    o[15] = cp.cohorts;
    if (c == 1) break;
  }
  *count = rows;
  return NMFK_OK;
}

namespace {
// Retire-aware schedule: position p of the new work list takes the unit at position perm[p] of the old one.  runs[] and
// state[] are copied into their other buffers (the old ones stay as they are: a snapshot copy may still read them), the
// unit's slot counts become those of the new geometry and both sum tables are folded into slot 0 in the order the
// consumers add the slots -- sum = ((s0 + s1) + ...) + 0 + ... is then bit for bit what it was.  One wave per unit.
__global__ __launch_bounds__(64) void replan_kernel(char *arena, const NmfkRun *runs_old, const NmfkState *state_old,
                                                    NmfkRun *runs_new, NmfkState *state_new, const int32_t *perm, int nsW,
                                                    int nsH, int PW, int PH, int PWz, int PHz,  // PW / PH: slots in use so far; PWz / PHz: slots to leave defined
                                                    int64_t opart0, int64_t opart_stride, int live) {  // partial-numerator buffer of position p < live (idle between half-steps)
  const int p = blockIdx.x, q = perm[p], t = threadIdx.x;
  NmfkRun rd = runs_old[q];
  if (t == 0) {
    state_new[p] = state_old[q];
    rd.nsW = nsW;
    rd.nsH = nsH;
    rd.opart = opart0 + (int64_t)(p < live ? p : 0) * opart_stride;  // (a stopped unit's kernels never run again)
    runs_new[p] = rd;
  }
  const int kp = rd.kp;
  if (t < kp) {
    double *tabs[2] = {(double *)(arena + rd.osumW), (double *)(arena + rd.osumH)};
    const int P[2] = {PW, PH}, Pz[2] = {PWz, PHz};
    for (int f = 0; f < 2; ++f) {
      double sd = 0;
      for (int pp = 0; pp < P[f]; ++pp) sd += tabs[f][pp * kp + t];
      tabs[f][t] = sd;
      for (int pp = 1; pp < Pz[f]; ++pp) tabs[f][pp * kp + t] = 0.0;
    }
  }
}
}  // namespace

NMFK_EXPORT int nmfk_mu_sweep(nmfk_ctx *ctx, int nk, const int32_t *ks, int nruns, const float *const *Winit,
                              const float *const *Hinit, const uint64_t *seeds, const nmfk_mu_params *params,
                              float *const *W_out, float *const *H_out, float *const *frob_out,
                              double *const *sse_out, int32_t *const *iters_out, int32_t *const *reason_out) {
  if (!ctx || !ks || !params || !W_out || !H_out || !frob_out) return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (!ctx->Xc && !ctx->sparse) return fail(NMFK_ERR_NO_X, "nmfk_set_X has not been called");
  if (nk <= 0 || nruns <= 0) return fail(NMFK_ERR_BAD_ARG, "nk and nruns must be positive");
  const nmfk_mu_params P = *params;
  if (P.compute != NMFK_COMPUTE_F32 && P.compute != NMFK_COMPUTE_F64) return fail(NMFK_ERR_BAD_ARG, "bad compute mode");
  if (P.maxiter < 0 || P.maxiter > 0x7ffffff0) return fail(NMFK_ERR_BAD_ARG, "bad maxiter");
  const int n = (int)ctx->n, m = (int)ctx->m;
  for (int q = 0; q < nk; ++q) {
    if (ks[q] < 1) return fail(NMFK_ERR_BAD_ARG, "k must be >= 1");
    if (ks[q] > NMFK_MAX_K) return fail(NMFK_ERR_UNSUPPORTED, "k exceeds NMFK_MAX_K (64)");
    const bool hasW = Winit && Winit[q], hasH = Hinit && Hinit[q];
    if ((!hasW || !hasH) && !seeds) return fail(NMFK_ERR_BAD_ARG, "seeds are required where Winit/Hinit are null");
    if (!W_out[q] || !H_out[q] || !frob_out[q]) return fail(NMFK_ERR_BAD_ARG, "null output pointer");
  }
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const bool f64 = P.compute == NMFK_COMPUTE_F64;
  const size_t tsz = f64 ? sizeof(double) : sizeof(float);
  const int nunits = nk * nruns;

  // ranks sorted by k descending (long units first; units with kp > 16 form a prefix)
  std::vector<int> order(nk);
  for (int q = 0; q < nk; ++q) order[q] = q;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ks[a] > ks[b]; });

  // Launch geometry of the two half-steps (see step_body): wsplit = 1 when the lane dimension alone fills the
  // chip, else the four waves of a workgroup share 64*LB lane elements and split the loop range; grid-level
  // splits S > 1 (finished by the reduce kernel) only when there are too few units to fill the chip otherwise.
  const Tuning T = read_tuning();
  const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  const int target = T.target_wgs > 0 ? T.target_wgs : 2 * cus;
  struct Geo {
    int wsplit, S, dchunk, fused, slots;
  };
  // ranks above 16 use the all-MFMA half-step (fp32 compute, no missing data, dense X)
  const bool mfma_ok = !f64 && !ctx->sparse && ctx->nan_count == 0 && n >= 16 && m >= 16;
  const bool wide_ok = T.wide && mfma_ok;
  auto use_wide_k = [&](int k) { return wide_ok && k > 16; };
  // ... and of those the widths whose first product runs from bf16 splits (wide2_step_kernel, nmfk_step_hyb.hip)
  const bool tile_fits = (int64_t)n * m * 4 < ((int64_t)1 << 32) - 4096;  // (buffer loads: 32-bit byte offsets into the tiled X)
  auto use_wide2_k = [&](int k) { return use_wide_k(k) && T.wide2 && tile_fits && nmfk_wide2_ok(nmfk_padded_k(k)); };
  // Ranks in [hyb_mink, 16]: split-operand MFMA half-step (nmfk_step_hyb.hip).  Its cost does not depend on the rank
  // and ONE instantiation serves all ranks, so its units share one launch group.  Two schedules use it:
  //  * few restarts per rank (<= 8; a rank's share at 4-8 GPUs): the ranks >= 6 as one group on it, the small ranks
  //    beside it on their per-rank packed-VALU launches; with <= 4 restarts every rank 2..16 joins the group and no
  //    packed-VALU launch is left;
  //  * many restarts per rank: a TWO-PHASE sweep -- the ranks >= k0 run FIRST, as one group with the GPU to themselves
  //    (its fp32 MFMAs and the packed FMAs of the other ranks' kernels share the multipliers, so the two kinds must not
  //    run side by side), then the other ranks on their per-rank packed-VALU launches.
  // The two-phase rule is a cost model in (n, m, ranks, restarts), constants measured on MI355X (profiles/r02/
  // schedule_shapes.txt): per factorization and iteration a packed-VALU unit costs E' * (1406 + 324 k) ns and a unit
  // of the group E' * 3670 ns + 8.5 ns per workgroup of its W half-step, E' = n*m / (8192*512); inside the mixed
  // sweep the packed-VALU ranks overlap better than one at a time, which moves the break-even from k = 7 to 8.8 at the
  // reference shape.  The phases are taken when (a) the group's launches are long enough not to be launch-bound,
  // (b) its workgroups in the short dimension cover at least half the CUs, and (c) the model promises >= 300
  // rank-restarts of saving (k = 2:12 x 32 would lose 3 %, k = 2:16 x 32 gains 15 %, x 16 gains 5 %).
  // Round 4: the rule used to ask for >= 16 restarts per rank, which left 9..15 -- the reference's README example runs 10 -- on the
  // per-rank packed-VALU launches: k = 2:16 x 9..15 at 8192 x 512 took 0.78-1.03 ms per iteration there against 0.51-0.77 ms on the
  // group, k = 2:5 x 10 0.23 against 0.14-0.18 (profiles/r04/schedule_9_to_15_restarts.txt); (b) now asks for a quarter of the CUs.
  int hyb_on = T.hyb, hyb_mink = T.hyb_mink;
  const int hyb_groups = T.hyb_groups;  // merged sweeps: number of mixed-rank launch groups of the split-operand MFMA kernel
  int merge = T.merge;
  const bool merge_env = T.merge >= 0;
  if (merge < 0) merge = nruns <= NMFK_MERGE_MAX_RUNS ? std::min(nruns, NMFK_MERGE_GROUPS) : 0;
  if (ctx->sparse) merge = 0;
  merge = std::min(merge, nruns);
  bool hyb_phases = false;
  // (the kernel's buffer loads address X with 32-bit byte offsets from the array base)
  const bool hyb_fits = mfma_ok && (int64_t)n * m * 4 < ((int64_t)1 << 32) - 4096;
  if (hyb_on < 0) {  // automatic (an explicit NMFK_MERGE keeps the packed-VALU groups)
    // Round 3: the matrix-pipe kernel no longer pads a rank to 16 signals (nmfk_step_hyb.hip: one bf16 MFMA and 4x4x1
    // numerator blocks for k <= 4, two and eight for k <= 8): first rank 2.  (Tuning::hyb_small = 0 is round 2's rule: first
    // rank = break-even against the 16-signal form, k0 above; 2 / 6 in merged sweeps.)
    const double Erel = (double)n * m / (8192.0 * 512.0);
    const double wg_ns = 8.5 * ((double)n + m) / 256.0;
    const double k0 = 8.8 + (wg_ns / Erel - 8.5 * 34.0) / 324.0;
    const int mk = hyb_mink >= 0 ? hyb_mink
                   : T.hyb_small ? 2
                                 : (nruns <= 4 ? 2 : nruns <= 8 ? 6 : std::min(16, (int)ceil(k0)));
    int hyb_units = 0, hyb_kmax = 0;
    for (int q = 0; q < nk; ++q) {
      if (hyb_fits && ks[q] <= 16 && ks[q] >= mk) {
        hyb_units += nruns;
        hyb_kmax = std::max(hyb_kmax, ks[q]);
      }
    }
    // Round 4: rounds 2-3 guarded the group with a cost model of THEIR kernels -- >= 16 restarts per rank (or <= 8: merged
    // sweeps), a launch of >= 50 us, workgroups for half the CUs, a break-even rank.  Measured again over eight shapes x ten
    // sweeps (profiles/r04/drivers/r4_schedule_probe.py, profiles/r04/schedule_probe_before.txt) the guards cost up to 4x: the group beats the
    // per-rank packed-VALU launches almost everywhere -- 9..15 restarts per rank (0.78-1.04 -> 0.47-0.75 ms per iteration at
    // 8192 x 512), small matrices (1024 x 128, k = 2:16 x 10: 0.279 -> 0.064 ms), single ranks (k = 16 x 10: 0.171 -> 0.107).
    // What is left on the packed-VALU kernels by measurement: a few units of the smallest ranks only (k = 3 x 10: 0.055 against
    // 0.063 ms).  (A second exception -- ranks <= 5 on a large matrix whose short W half-step loop could not take the resident form --
    // went away when that form learnt loop lengths that are not multiples of 64: 20000 x 1000, k = 4 x 64: 1.05 -> 0.63 ms against
    // 0.72 on the packed-VALU kernel; one measured case is left where that kernel is ahead, k = 2:5 x 10 there: 0.55 against 0.70.)
    const bool tiny_small = hyb_kmax <= 4 && hyb_units < 24;
    if (hyb_units > 0 && !merge_env && !ctx->sparse && !tiny_small) {
      hyb_on = 1;
      // the other ranks run BEHIND the group (the group's fp32 MFMAs and their kernels' packed FMAs / fp32 MFMAs share the
      // multipliers; side by side k = 2:32 x 8 took 1.62 ms per iteration at 8192 x 512 against 1.43 phased, 1.77 against
      // 0.57 at 512 x 8192), each phase with its own launch geometry
      hyb_phases = true;  // (one mixed-rank launch group for the ranks on it; the other ranks in phase 1)
      bool low_left = false;  // ranks below the group's first one (only with Tuning::hyb_small = 0 or NMFK_HYB_MINK)
      for (int q = 0; q < nk; ++q) low_left = low_left || ks[q] < mk;
      merge = (low_left && (merge > 0 || nruns <= 8)) ? 1 : 0;  // (few restarts: they share one packed-VALU group beside the matrix-pipe group)
    } else {
      hyb_on = 0;
    }
    hyb_mink = mk;
  }
  if (T.phases >= 0) hyb_phases = hyb_on && T.phases != 0;
  if (hyb_mink < 0) hyb_mink = 5;
  // The ranks <= 16 that are not on the MFMA group share `merge` mixed-rank packed-VALU launch groups (step_kernel_multi)
  // when the sweep has few restarts per rank.  (Round 2 met the gfx950 packed-fp32 hazard in this kernel first -- DESIGN.md,
  // "Known hazard"; its generated code is free of the unsafe instruction form now and checked by tests/test_isa_lint.py.)
  const bool valu_merged = merge > 0 && (f64 || NMFK_WITH_MERGED_F32 != 0);
  auto use_hyb_k = [&](int k) { return hyb_on && hyb_fits && k <= 16 && k >= hyb_mink; };
  // kernel variant of a rank on the split-operand MFMA half-step (nmfk_hyb_variant: 4 / 8 / 12 / 16)
  auto hyb_variant_of = [&](int k) { return T.hyb_small ? nmfk_hyb_variant(k) : 16; };
  int hyb_vmax = 4;  // widest variant among the units of the matrix-pipe groups (sizes their LDS)
  for (int q = 0; q < nk; ++q)
    if (use_hyb_k(ks[q])) hyb_vmax = std::max(hyb_vmax, hyb_variant_of(ks[q]));
  // lane elements per workgroup (= per sum-table slot) of the half-step kernel a rank runs
  auto lane_tile = [&](int k, int ws) {
    if (ctx->sparse) return NMFK_TILE;
    if (use_wide2_k(k) && ws == 1) return nmfk_wide2_lane_tile();
    if (use_wide_k(k)) return nmfk_mfma_wide_lane_tile(ws);
    if (use_hyb_k(k)) return nmfk_hyb_lane_tile(ws);
    if (valu_merged && k <= NMFK_MULTI_MAXK && !use_hyb_k(k)) return (ws > 1 ? 64 : NMFK_TILE) * NMFK_MULTI_LB;
    return (ws > 1 ? 64 : NMFK_TILE) * NMFK_LB_OF(nmfk_padded_k(k));
  };
  // Resident form of the split-operand MFMA half-step (nmfk_step_hyb.hip, hyb_res_kernel): when the loop dimension is
  // short enough for the whole loop factor to sit in LDS (the W half-step of a tall X), the units of the matrix-pipe
  // groups run it with res_wgs[which] workgroups of 16 waves per unit, each wave walking several pairs of lane tiles.
  HybMix hyb_mix;  // the units on the matrix-pipe kernels, by kernel variant
  for (int q = 0; q < nk; ++q)
    if (use_hyb_k(ks[q])) {
      const int v = hyb_variant_of(ks[q]);
      (v <= 4 ? hyb_mix.n4 : v <= 8 ? hyb_mix.n8 : hyb_mix.n16) += nruns;
    }
  // their launch geometry (plan_hyb_group: the cost model) -- the resident form's workgroups per unit, wsplit, S of both half-steps
  const HybCohortPlan hyb_cplan0 = plan_hyb_cohorts(n, m, cus, hyb_mix, T, hyb_on && T.hyb_res);
  const HybPlan &hyb_plan0 = hyb_cplan0.plan;
  int hyb_ranks = 0;
  for (int q = 0; q < nk; ++q) hyb_ranks += use_hyb_k(ks[q]) ? 1 : 0;
  if (T.debug && hyb_mix.units() > 0)
    fprintf(stderr, "[nmfk] plan of %d matrix-pipe units (%d / %d / %d of variant 16 / 8 / 4) at %d x %d: H res %d wsplit %d S %d (model %.1f us, busy %.2f) | W res %d wsplit %d S %d (%.1f us, %.2f) | cohorts %d\n",
            hyb_mix.units(), hyb_mix.n16, hyb_mix.n8, hyb_mix.n4, n, m, hyb_plan0.res[0], hyb_plan0.wsplit[0], hyb_plan0.S[0], hyb_plan0.us[0], hyb_plan0.busy[0],
            hyb_plan0.res[1], hyb_plan0.wsplit[1], hyb_plan0.S[1], hyb_plan0.us[1], hyb_plan0.busy[1], hyb_cplan0.cohorts);
  int res_wgs[2] = {0, 0};  // [0] H half-step (L = m, D = n), [1] W half-step (L = n, D = m)
  if (hyb_mix.units() > 0) res_wgs[0] = hyb_plan0.res[0], res_wgs[1] = hyb_plan0.res[1];
  // workgroups (= lane tiles = sum-table slots) of one unit of rank k in the half-step `which`
  auto tiles_of_res = [&](int k, int which, int L, int ws, const int (&res)[2]) {
    if (use_hyb_k(k) && res[which] > 0) return res[which];
    return (L + lane_tile(k, ws) - 1) / lane_tile(k, ws);
  };
  auto tiles_of = [&](int k, int which, int L, int ws) { return tiles_of_res(k, which, L, ws, res_wgs); };
  const int max_ws = T.max_wsplit;
  // phases of a two-phase sweep run one after the other, so each gets the geometry that fills the chip with ITS units
  // (Tuning::merge_phased, measured and left off: a merged sweep runs its matrix-pipe groups first and the mixed-rank packed-VALU group
  // behind them instead of side by side (slower: 140 vs 117 ms per 400 iterations at 4 restarts per rank).
  bool any_hyb_k = false;
  for (int q = 0; q < nk; ++q) any_hyb_k = any_hyb_k || use_hyb_k(ks[q]);
  const bool phased = hyb_phases || (valu_merged && any_hyb_k && T.merge_phased);
  auto phase_of_k = [&](int k) { return phased && !use_hyb_k(k) && !(merge > 0 && use_wide_k(k)) ? 1 : 0; };
  // units_of_rank < 0: the sweep as given (nruns units per rank); >= 0: that many units per rank (the tiers of the
  // retire-aware schedule plan a sweep that has shrunk), with `res` the resident-form plan of that sweep
  auto geometry_for = [&](int L, int D, int phase, int which, const int (&res)[2], double units_of_rank) {
    Geo g;
    {  // a phase that holds the matrix-pipe units and nothing else (the default schedule's phase 0): their plan
      int ranks_in = 0, hyb_in = 0;
      for (int q = 0; q < nk; ++q)
        if (phase_of_k(ks[q]) == phase) {
          ++ranks_in;
          hyb_in += use_hyb_k(ks[q]) ? 1 : 0;
        }
      if (ranks_in > 0 && hyb_in == ranks_in && hyb_in == hyb_ranks && units_of_rank < 0 && hyb_plan0.res[which] == res[which]) {
        const HybPlan &p = hyb_plan0;
        return Geo{p.wsplit[which], p.S[which], p.dchunk[which], p.fused[which], p.slots[which]};
      }
    }
    const double per_rank = units_of_rank >= 0 ? units_of_rank : (double)nruns;
    auto wgs = [&](int ws) {  // workgroups of one half-step over all units of the phase
      double t = 0;
      for (int q = 0; q < nk; ++q)
        if (phase_of_k(ks[q]) == phase) t += (double)tiles_of_res(ks[q], which, L, ws, res) * per_rank;
      return std::max<int64_t>((int64_t)ceil(t - 1e-9), 1);
    };
    g.wsplit = 1;
    // the waves of a workgroup split the loop range (and stage for themselves) when whole workgroups would not fill the
    // chip: below 2 per CU for the packed-VALU kernels, below 1.5 per CU when the phase runs the split-operand MFMA
    // kernel, whose per-wave staging form is the slower one (240 units x 2 lane tiles: 0.376 ms shared vs 0.436 ms split)
    bool phase_hyb = false;
    for (int q = 0; q < nk; ++q) phase_hyb = phase_hyb || (phase_of_k(ks[q]) == phase && use_hyb_k(ks[q]));
    const int target_ws = (T.target_wgs > 0 || !phase_hyb) ? target : hyb_target_ws(cus);
    // (the split-operand wide-rank kernel has no form in which the waves of a workgroup split the loop range: a phase
    //  that runs it fills the chip by splitting the range over workgroups instead, S below)
    bool phase_wide2 = false;
    for (int q = 0; q < nk; ++q) phase_wide2 = phase_wide2 || (phase_of_k(ks[q]) == phase && use_wide2_k(ks[q]));
    if (wgs(1) < target_ws && !phase_wide2) g.wsplit = (max_ws >= 8 && D >= 8 * 64) ? 8 : 4;
    const int64_t have = wgs(g.wsplit);
    int S = (int)((target + have - 1) / have);
    const int maxS = std::max(1, D / (64 * g.wsplit));
    g.S = std::max(1, std::min(S, maxS));
    g.dchunk = (D + g.S - 1) / g.S;
    if (g.S > 1) {  // split points on multiples of 16 (the MFMA kernels read X in 16-step blocks)
      g.dchunk = (g.dchunk + 15) & ~15;
      g.S = (D + g.dchunk - 1) / g.dchunk;
    }
    g.fused = g.S == 1;
    g.slots = 1;  // slots of the sum tables = the most lane tiles any rank's kernel uses
    for (int q = 0; q < nk; ++q)
      if (phase_of_k(ks[q]) == phase)
        g.slots = std::max(g.slots, tiles_of_res(ks[q], which, L, g.wsplit, res));
    // not fused: reduce_kernel finishes the half-step, one workgroup per (slot, unit) -- with the few lane tiles that made
    // the split necessary that was 8 workgroups per unit walking 16 K elements x S partials each (345 us per launch at
    // 65536 x 2048, k = 64, 8 units, against a 1.8 ms half-step): any partition of the lane range into slots is valid,
    // so give it up to 64 slots of >= 32 lane elements
    if (!g.fused) g.slots = std::max(g.slots, std::min(64, (L + 31) / 32));
    return g;
  };
  auto geometry = [&](int L, int D, int phase, int which) { return geometry_for(L, D, phase, which, res_wgs, -1.0); };
  Geo ghp[2] = {geometry(m, n, 0, 0), geometry(m, n, phased ? 1 : 0, 0)};
  Geo gwp[2] = {geometry(n, m, 0, 1), geometry(n, m, phased ? 1 : 0, 1)};
  // sparse X, ranks up to 32: blocked form (a lane element per thread, the gathered factor through LDS) when the sliced ELL of
  // the orientation exists (nmfk_set_X_csc) and the launch fills the GPU (1024 lane elements per workgroup: the H half-step
  // of a matrix with few columns does not, the gather form serves it)
  bool sp_blk[2] = {false, false};  // [0]: H half-step, [1]: W half-step
  if (ctx->sparse) {  // gather kernels, always finished in-kernel; the H half-step's slots are per pass (see NmfkSparseArgs)
    int slots_h = 1;
    for (int q = 0; q < nk; ++q) {
      const int sl = nmfk_sp_slot(nmfk_padded_k(ks[q]), 1);
      slots_h = std::max(slots_h, (m + sl - 1) / sl);
    }
    int64_t blk_units = 0;
    for (int q = 0; q < nk; ++q) blk_units += nmfk_sp_blk_rank(nmfk_padded_k(ks[q])) ? nruns : 0;
    const int64_t cus = ctx->prop.multiProcessorCount;
    const int64_t fill = T.sp_blk >= 2 ? 0 : cus / 2;  // (NMFK_SP_BLK=2: whatever the size -- the tests' small cases)
    sp_blk[0] = T.sp_blk && ctx->ell[1] && tsz == 4 && blk_units * ((m + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS) >= fill;
    sp_blk[1] = T.sp_blk && ctx->ell[0] && tsz == 4 && blk_units * ((n + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS) >= fill;
    ghp[0] = ghp[1] = Geo{1, 1, n, 1, slots_h};
    gwp[0] = gwp[1] = Geo{1, 1, m, 1, (n + NMFK_TILE - 1) / NMFK_TILE};
  }
  // Retire-aware schedule (round 4).  The reference's loop guard (Mult:64) ends every restart on its own, and on structured
  // data the restarts of a sweep stop anywhere between a few hundred iterations and maxiter; a launch geometry chosen for
  // all units of the sweep then runs a shrinking set of them on a fraction of the chip (H half-step: two workgroups per
  // unit, whatever is left).  When every unit of the sweep sits in ONE launch group on the matrix-pipe kernels (dense fp32,
  // ranks 2..16 -- the bench and the usual `execute` call), the sweep is planned as TIERS: tier j is the geometry for
  // ceil(units / 2^j) units.  At a check at which the units still active (as of the previous check's snapshot) fit the next
  // tier, the host -- in stream order, without waiting for the GPU -- moves the active units to the front of the work list
  // (replan_kernel: runs[] and state[] are copied in the new order into their second buffers, the sum tables folded into
  // slot 0, which leaves every sum the next half-step forms bit for bit what it was) and switches to the tier's geometry.
  // A unit's results are a deterministic function of the sweep as before (the decisions hang on the unit states at fixed
  // iterations); they differ from a run without re-planning in the last bits only, as they do between launch geometries.
  struct Tier {
    int count;
    Geo gh, gw;
    int res[2];
    int nsH, nsW;  // slots a unit's kernels write under the tier (NmfkRun::nsH / nsW)
    int PH, PW;    // slots the tier's helper kernels (reduce, clamp) cover: written or zeroed
    int ncoh;      // cohorts the tier's units run as
  };
  std::vector<Tier> tiers;
  {
    bool all_hyb = nk > 0;
    for (int q = 0; q < nk; ++q) all_hyb = all_hyb && use_hyb_k(ks[q]) && ks[q] <= NMFK_MULTI_MAXK;
    const bool one_group = all_hyb && !ctx->sparse && !f64 && (merge > 0 || hyb_phases) && hyb_groups == 1;
    if (T.replan && one_group && (nunits >= 32 || T.replan >= 2)) {
      tiers.push_back({nunits, ghp[0], gwp[0], {res_wgs[0], res_wgs[1]}, 0, 0, 0, 0, hyb_cplan0.cohorts});
      {  // the standalone rule (plan_hyb_group: what the later tiers and the CPU tests use) is the general one for such a sweep
        const HybPlan &p0 = hyb_plan0;
        const Geo g0[2] = {ghp[0], gwp[0]};
        bool same = true;
        for (int w = 0; w < 2; ++w)
          same = same && p0.res[w] == res_wgs[w] && p0.wsplit[w] == g0[w].wsplit && p0.S[w] == g0[w].S && p0.dchunk[w] == g0[w].dchunk &&
                 p0.fused[w] == g0[w].fused && p0.slots[w] == g0[w].slots;
        if (!same) {  // (an unusual NMFK_TARGET_WGS or shape: a pure optimisation must not fail the sweep -- static schedule)
          if (T.debug) fprintf(stderr, "[nmfk] the tier planner disagrees with the sweep's launch geometry: static schedule\n");
          tiers.clear();
        }
      }
      for (int c = (nunits + 1) / 2; !tiers.empty() && c >= 1 && c < tiers.back().count; c = (c + 1) / 2) {
        const HybCohortPlan cp = plan_hyb_cohorts(n, m, cus, hyb_mix.scaled(c), T, T.hyb_res != 0);
        const HybPlan &p = cp.plan;
        Tier t;
        t.count = c;
        t.ncoh = cp.cohorts;
        t.res[0] = p.res[0], t.res[1] = p.res[1];
        t.gh = Geo{p.wsplit[0], p.S[0], p.dchunk[0], p.fused[0], p.slots[0]};
        t.gw = Geo{p.wsplit[1], p.S[1], p.dchunk[1], p.fused[1], p.slots[1]};
        t.nsH = t.nsW = t.PH = t.PW = 0;
        tiers.push_back(t);
        if (c == 1) break;
      }
    }
  }
  int Sh = std::max(ghp[0].S, ghp[1].S), Sw = std::max(gwp[0].S, gwp[1].S);  // (sizes the partial-numerator buffers)
  // slots of the sum tables (rowsum(H) is produced by the H half-step): one table size for all units
  const int PH = std::max(ghp[0].slots, ghp[1].slots), PW = std::max(gwp[0].slots, gwp[1].slots);
  int PHmax = PH, PWmax = PW;  // (the tables are allocated for the widest tier; a tier's kernels see its own PH / PW)
  for (Tier &t : tiers) {  // (the resident form finishes itself: no partial numerators)
    Sh = std::max(Sh, t.res[0] > 0 ? 1 : t.gh.S);
    Sw = std::max(Sw, t.res[1] > 0 ? 1 : t.gw.S);
    t.PH = &t == &tiers[0] ? PH : t.gh.slots;
    t.PW = &t == &tiers[0] ? PW : t.gw.slots;
    t.nsH = (t.gh.fused || t.res[0] > 0) ? tiles_of_res(ks[0], 0, m, t.gh.wsplit, t.res) : t.PH;
    t.nsW = (t.gw.fused || t.res[1] > 0) ? tiles_of_res(ks[0], 1, n, t.gw.wsplit, t.res) : t.PW;
    PHmax = std::max(PHmax, t.PH);
    PWmax = std::max(PWmax, t.PW);
  }
  const int tiles_n = (n + NMFK_TILE - 1) / NMFK_TILE;  // objective kernel tiles

  // arena layout
  Bump B;
  const size_t o_runs = B.take(sizeof(NmfkRun) * nunits);
  const size_t o_state = B.take(sizeof(NmfkState) * nunits);
  const size_t o_flag = B.take(256);
  const size_t o_args = B.take(4 * sizeof(NmfkStepArgs));
  // retire-aware schedule: the second buffers of runs[] / state[], the permutation and the argument blocks of every re-plan
  const bool replanning = tiers.size() > 1;
  const size_t o_runs2 = replanning ? B.take(sizeof(NmfkRun) * nunits) : 0;
  const size_t o_state2 = replanning ? B.take(sizeof(NmfkState) * nunits) : 0;
  const size_t o_perm = replanning ? B.take(sizeof(int32_t) * (size_t)nunits * tiers.size()) : 0;
  const size_t o_args2 = replanning ? B.take(2 * sizeof(NmfkStepArgs) * tiers.size()) : 0;
  const size_t o_ptrs = B.take(sizeof(void *) * 7 * nk);
  const int trace_stride = ctx->trace_objective ? (int)std::max<int64_t>(1, P.maxiter / 10) : 0;
  const size_t o_trace = ctx->trace_objective ? B.take(sizeof(double) * (size_t)nunits * trace_stride) : 0;
  // Partial numerators (half-steps whose loop range is split over workgroups) and check_b's index scratch.  Static schedule: a buffer
  // per unit.  Retire-aware schedule: the late tiers run few units with many splits (S up to 64), so a buffer per unit sized for the
  // widest tier would cost S_max * kp * max(n, m) for EVERY unit of the sweep (~1 GB at 8192 x 512 x 480); the buffers are idle between
  // half-steps, so they belong to the POSITIONS of the work list instead -- position p of tier t at p * stride_t of one pool sized
  // for the largest count_t * stride_t (count_t halves where S_t doubles); replan_kernel hands them out.
  size_t part_stride0 = 0, o_partpool = 0;
  std::vector<size_t> part_stride(tiers.size(), 0);
  if (replanning) {
    int kpm = 1;
    for (int q = 0; q < nk; ++q) kpm = std::max(kpm, nmfk_padded_k(ks[q]));
    size_t pool = 0;
    for (size_t j = 0; j < tiers.size(); ++j) {
      const Tier &t = tiers[j];
      const size_t sh = (t.res[0] > 0 || t.gh.fused) ? 0 : (size_t)t.gh.S, sw = (t.res[1] > 0 || t.gw.fused) ? 0 : (size_t)t.gw.S;
      size_t b = std::max(tsz * std::max(sh * kpm * m, sw * kpm * n), sizeof(int32_t) * (size_t)m);
      b = (b + 255) & ~(size_t)255;
      part_stride[j] = b;
      pool = std::max(pool, b * (size_t)t.count);
    }
    part_stride0 = part_stride[0];
    o_partpool = B.take(pool);
  }
  std::vector<NmfkRun> runs(nunits);
  std::vector<size_t> o_Wi(nk, 0), o_Hi(nk, 0), o_Wo(nk), o_Ho(nk), o_frob(nk), o_iters(nk), o_reason(nk);
  struct Group {
    int k, kp, begin, count;
    int hyb;    // split width of the split-operand MFMA kernel (8 / 16) when the group runs on it, else 0
    int phase;  // groups of phase 0 run to the end before those of phase 1 start
  };
  // Launch groups = contiguous unit ranges.  Default: one group per rank (units sorted by k descending), each with its
  // own kernel instantiation and stream.  With few restarts per rank the per-rank launches are tiny and the loop is
  // launch-bound: then the ranks <= 16 are merged into `merge` super-groups (restart r of every rank goes to
  // super-group r mod merge; kp = 0 marks a mixed-rank group, served by step_kernel_multi).
  std::vector<Group> groups;
  std::vector<std::pair<int, int>> ulist;  // (index into ks, restart) in unit order
  ulist.reserve(nunits);
  for (int oi = 0; oi < nk; ++oi) {
    const int q = order[oi], k = ks[q];
    if (merge > 0 && k <= NMFK_MULTI_MAXK && (use_hyb_k(k) || valu_merged)) continue;
    if (hyb_phases && use_hyb_k(k)) continue;
    // sparse X: one kernel instantiation serves every rank with the same number of lanes per lane element, so those
    // ranks share a launch group (31 per-rank launches of 16 units each left the GPU half empty: 33 -> 21 ms)
    // ranks above 16 on the split-operand kernel: the instantiation is chosen by the padded width's blocks of sixteen signals
    // (32 / 48 / 64) and a workgroup takes its unit's rank from NmfkRun, so the ranks of one instantiation share a launch group
    // (round 4; NMFK_WIDE_GROUPS=0: a group per rank.  k = 2:40 x 4 at 8192 x 512: 1.51 -> 1.27 ms per iteration, k = 17:32 x 2:
    //  0.41 -> 0.31, at 1024 x 128 0.47 -> 0.18 -- profiles/r04/wide_rank_groups.txt)
    auto wide2_nb = [](int kp) { return kp <= 32 ? 2 : kp <= 48 ? 3 : 4; };
    const bool w2 = T.wide_groups && !ctx->sparse && !f64 && use_wide_k(k) && use_wide2_k(k) && ghp[phase_of_k(k)].wsplit == 1 &&
                    gwp[phase_of_k(k)].wsplit == 1;
    // ... as long as the group is short of workgroups: a launch group per rank keeps the half-steps of different ranks overlapping
    // (no launch-wide barrier between them), which is worth 4-10 % once a rank fills the chip by itself: ranks join a group until
    // its W half-step has T.wide_groups (2) workgroups per CU
    const int64_t w2_wgs = groups.empty() ? 0 : (int64_t)groups.back().count * ((n + nmfk_wide2_lane_tile() - 1) / nmfk_wide2_lane_tile());
    if (w2 && !groups.empty() && groups.back().hyb == 0 && groups.back().kp > 16 && use_wide2_k(groups.back().k) &&
        groups.back().phase == phase_of_k(k) && wide2_nb(groups.back().kp) == wide2_nb(nmfk_padded_k(k)) &&
        w2_wgs < (int64_t)T.wide_groups * cus) {
      groups.back().count += nruns;  // (ranks come in descending order: k and kp of the group stay its widest rank's)
    } else if (ctx->sparse && !groups.empty() && nmfk_sp_lpr(groups.back().kp) == nmfk_sp_lpr(nmfk_padded_k(k))) {
      groups.back().kp = std::max(groups.back().kp, nmfk_padded_k(k));
      groups.back().count += nruns;
    } else {
      groups.push_back({k, nmfk_padded_k(k), (int)ulist.size(), nruns, use_hyb_k(k) ? hyb_vmax : 0, phase_of_k(k)});
    }
    for (int r = 0; r < nruns; ++r) ulist.push_back({q, r});
  }
  // merged sweeps: the ranks of the split-operand MFMA kernel (its cost does not depend on the rank, one instantiation
  // serves them all at split width 16) form mixed-rank groups of their own, the other ranks <= 16 the VALU ones
  const int hg = (merge > 0 || hyb_phases) ? std::min(hyb_groups, nruns) : 0;
  // (ONE launch group whatever the kernel variants of its units are: the kernels switch per workgroup, NmfkRun::hyb)
  for (int g = 0; g < hg; ++g) {
    Group G{0, 0, (int)ulist.size(), 0, hyb_vmax, 0};
    for (int oi = 0; oi < nk; ++oi) {
      const int q = order[oi];
      if (ks[q] > NMFK_MULTI_MAXK || !use_hyb_k(ks[q])) continue;
      for (int r = g; r < nruns; r += hg, ++G.count) ulist.push_back({q, r});
    }
    if (G.count > 0) groups.push_back(G);
  }
  for (int g = 0; g < (valu_merged ? merge : 0); ++g) {
    Group G{0, 0, (int)ulist.size(), 0, 0, phased ? 1 : 0};
    for (int oi = 0; oi < nk; ++oi) {
      const int q = order[oi];
      if (ks[q] > NMFK_MULTI_MAXK || use_hyb_k(ks[q])) continue;
      for (int r = g; r < nruns; r += merge, ++G.count) ulist.push_back({q, r});
    }
    if (G.count > 0) groups.push_back(G);
  }
  // Cohorts (round 5).  A half-step launch of few units ends in a tail -- its last workgroups leave most CUs idle -- and the next
  // launch of the same units cannot start before it has drained: the kernel trace of a 60-unit share of the bench sweep shows no
  // gaps between launches, the time is inside them (H / W half-step 125 / 108 us where an eighth of the 480-unit launches is
  // 84 / 89).  So the units of the matrix-pipe group are dealt round-robin (the list is sorted by rank: every cohort gets the
  // same mix) to `ncoh` COHORTS, contiguous in the work list, each with its own stream: their launches share the launch
  // geometry of the group (a unit's arithmetic, and therefore its bits, does not depend on its cohort) and overlap freely,
  // one cohort's W half-step filling the CUs another's H half-step leaves idle.  The check block runs per cohort; the
  // retire-aware schedule joins the cohort streams for a re-plan and deals the units still active out again.
  std::vector<std::vector<std::pair<int, int>>> coh(groups.size());  // [group][cohort] = (first unit, units)
  for (size_t j = 0; j < groups.size(); ++j) {
    const Group &G = groups[j];
    int nc = 1;
    if (G.hyb && G.kp == 0 && !ctx->sparse && !f64) nc = G.count == hyb_mix.units() ? hyb_cplan0.cohorts : (T.cohorts > 0 ? T.cohorts : 1);
    nc = std::max(1, std::min(nc, G.count));
    if (nc > 1) {
      std::vector<std::pair<int, int>> seg(ulist.begin() + G.begin, ulist.begin() + G.begin + G.count);
      int w = G.begin;
      for (int c = 0; c < nc; ++c) {
        const int b = w;
        for (int i = c; i < G.count; i += nc) ulist[(size_t)w++] = seg[(size_t)i];
        coh[j].push_back({b, w - b});
      }
    } else {
      coh[j].push_back({G.begin, G.count});
    }
  }
  // objective partials per unit: a workgroup of 256 rows each (sse kernels), or, when the check is deferred into the next H
  // half-step (see the loop), one per workgroup of that launch -- in any tier of the retire-aware schedule
  int obj_cap = tiles_n + 1;
  {
    auto hparts = [&](int res, int wsplit, int S) { return res > 0 ? res : (m + nmfk_hyb_lane_tile(wsplit) - 1) / nmfk_hyb_lane_tile(wsplit) * S; };
    int want = 0;
    for (const Group &G : groups)
      if (G.hyb || (use_wide_k(G.k) && use_wide2_k(G.k))) want = std::max(want, hparts(G.hyb ? res_wgs[0] : 0, ghp[G.phase].wsplit, ghp[G.phase].S));
    for (const Tier &t : tiers) want = std::max(want, hparts(t.res[0], t.gh.wsplit, t.gh.S));
    if (want <= 8192) obj_cap = std::max(obj_cap, want);
  }
  {
    for (int u = 0; u < nunits; ++u) {
      const int q = ulist[u].first, r = ulist[u].second, k = ks[q], kp = nmfk_padded_k(k);
      {
        NmfkRun &rd = runs[u];
        rd.k = k;
        rd.kp = kp;
        rd.kidx = q;
        rd.ridx = r;
        rd.oWt = (int64_t)B.take(tsz * (size_t)kp * n);
        rd.oH0 = (int64_t)B.take(tsz * (size_t)kp * m);
        rd.oH1 = P.Hfixed ? rd.oH0 : (int64_t)B.take(tsz * (size_t)kp * m);
        const size_t pe = std::max((size_t)Sh * kp * m, (size_t)Sw * kp * n);
        rd.opart = replanning ? (int64_t)(o_partpool + (size_t)u * part_stride0) : (int64_t)B.take(std::max(tsz * pe, sizeof(int32_t) * (size_t)m));
        rd.osumW = (int64_t)B.take(sizeof(double) * (size_t)PWmax * kp);
        rd.osumH = (int64_t)B.take(sizeof(double) * (size_t)PHmax * kp);
        rd.ossepart = (int64_t)B.take(sizeof(double) * (size_t)obj_cap);  // (sparse objective: slot 0 = <W'W, HH'>)
        rd.ocanon = (int64_t)B.take(sizeof(int32_t) * 2 * (size_t)m);  // (the partition, then check_b's index scratch)
        rd.ogram = ctx->sparse ? (int64_t)B.take(sizeof(double) * nmfk_gram_doubles(n, m, kp)) : 0;
        rd.osnapW = (int64_t)B.take(sizeof(float) * 16);
        rd.seed = seeds ? seeds[(size_t)q * nruns + r] : 0;
        // slots the unit's kernels write: the fused half-step one per lane tile, the grid-parallel helpers any count
        const Geo &gh = ghp[phase_of_k(k)], &gw = gwp[phase_of_k(k)];
        rd.nsH = (gh.fused || (use_hyb_k(k) && res_wgs[0] > 0)) ? tiles_of(k, 0, m, gh.wsplit) : PH;
        if (ctx->sparse)  // (a slot per pass of 256 / LPR columns in the gather form, per 1024 columns in the blocked form)
          rd.nsH = sp_blk[0] && nmfk_sp_blk_rank(kp) ? (m + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS : (m + nmfk_sp_slot(kp, 1) - 1) / nmfk_sp_slot(kp, 1);
        rd.nsW = (gw.fused || (use_hyb_k(k) && res_wgs[1] > 0)) ? tiles_of(k, 1, n, gw.wsplit) : PW;
        if (ctx->sparse)  // (256 rows per slot in the gather form, 1024 in the blocked form)
          rd.nsW = sp_blk[1] && nmfk_sp_blk_rank(kp) ? (n + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS : gw.slots;
        rd.hyb = 0;
        rd.uid = u;
        if (use_hyb_k(k)) rd.hyb = hyb_variant_of(k);
      }
    }
  }
  for (int q = 0; q < nk; ++q) {
    const size_t k = (size_t)ks[q];
    if (Winit && Winit[q]) o_Wi[q] = B.take(sizeof(float) * nruns * k * n);
    if (Hinit && Hinit[q]) o_Hi[q] = B.take(sizeof(float) * nruns * k * m);
    o_Wo[q] = B.take(sizeof(float) * nruns * k * n);
    o_Ho[q] = B.take(sizeof(float) * nruns * k * m);
    o_frob[q] = B.take(sizeof(float) * nruns);
    o_iters[q] = B.take(sizeof(int32_t) * nruns);
    o_reason[q] = B.take(sizeof(int32_t) * nruns);
  }
  if (ctx->arena.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory (sweep arena)");
  char *A = ctx->arena.p;
  if (ensure_pinned(ctx, 2 * sizeof(NmfkState) * (size_t)nunits + 4096)) return fail(NMFK_ERR_HIP, "hipHostMalloc failed");

  // pointer tables: [Winit | Hinit | Wout | Hout | frob | iters | reason] x nk
  std::vector<void *> ptrs(7 * (size_t)nk, nullptr);
  for (int q = 0; q < nk; ++q) {
    ptrs[0 * nk + q] = (Winit && Winit[q]) ? (void *)(A + o_Wi[q]) : nullptr;
    ptrs[1 * nk + q] = (Hinit && Hinit[q]) ? (void *)(A + o_Hi[q]) : nullptr;
    ptrs[2 * nk + q] = A + o_Wo[q];
    ptrs[3 * nk + q] = A + o_Ho[q];
    ptrs[4 * nk + q] = A + o_frob[q];
    ptrs[5 * nk + q] = A + o_iters[q];
    ptrs[6 * nk + q] = A + o_reason[q];
  }
  HIPCHECK(hipMemcpyAsync(A + o_runs, runs.data(), sizeof(NmfkRun) * nunits, hipMemcpyHostToDevice, st));
  HIPCHECK(hipMemcpyAsync(A + o_ptrs, ptrs.data(), sizeof(void *) * ptrs.size(), hipMemcpyHostToDevice, st));
  HIPCHECK(hipMemsetAsync(A + o_flag, 0, 256, st));
  if (trace_stride) HIPCHECK(hipMemsetAsync(A + o_trace, 0xff, sizeof(double) * (size_t)nunits * trace_stride, st));  // NaN = no check
  for (int q = 0; q < nk; ++q) {
    const size_t k = (size_t)ks[q];
    if (Winit && Winit[q])
      HIPCHECK(hipMemcpyAsync(A + o_Wi[q], Winit[q], sizeof(float) * nruns * k * n, hipMemcpyDefault, st));
    if (Hinit && Hinit[q])
      HIPCHECK(hipMemcpyAsync(A + o_Hi[q], Hinit[q], sizeof(float) * nruns * k * m, hipMemcpyDefault, st));
  }
  // the host vectors above must outlive the async copies
  HIPCHECK(hipStreamSynchronize(st));

  NmfkRun *d_runs_buf[2] = {(NmfkRun *)(A + o_runs), replanning ? (NmfkRun *)(A + o_runs2) : nullptr};
  NmfkState *d_state_buf[2] = {(NmfkState *)(A + o_state), replanning ? (NmfkState *)(A + o_state2) : nullptr};
  const NmfkRun *d_runs = d_runs_buf[0];
  NmfkState *d_state = d_state_buf[0];
  void **d_ptrs = (void **)(A + o_ptrs);

  Sampler prof(ctx);

  NmfkInitArgs ia;
  ia.arena = A;
  ia.n = n;
  ia.m = m;
  ia.runs = d_runs;
  ia.state = d_state;
  ia.nunits = nunits;
  ia.Winit = (const float *const *)(d_ptrs + 0 * nk);
  ia.Hinit = (const float *const *)(d_ptrs + 1 * nk);
  ia.PW = PW;
  ia.PH = PH;
  ia.nan_flag = (int32_t *)(A + o_flag);
  if (f64)
    nmfk_launch_init_f64(ia, st);
  else
    nmfk_launch_init_f32(ia, st);
  bool any_hyb = false;
  for (int q = 0; q < nk; ++q) any_hyb = any_hyb || use_hyb_k(ks[q]) || use_wide2_k(ks[q]);  // (kernels that read the tiled X)
  const size_t tile_h = (size_t)((m + 15) / 16) * ((n + 15) / 16) * 256, tile_w = tile_h;  // floats
  if (any_hyb) {
    if (ctx->xtile_gen != ctx->xgen) {
      if (ctx->xtile.ensure(sizeof(float) * (tile_h + tile_w))) return fail(NMFK_ERR_HIP, "out of device memory (tiled X)");
      nmfk_launch_hyb_tile(ctx->Xc, m, n, (float *)ctx->xtile.p, st);           // H half-step: lanes = columns
      nmfk_launch_hyb_tile(ctx->Xr, n, m, (float *)ctx->xtile.p + tile_h, st);  // W half-step: lanes = rows
      ctx->xtile_gen = ctx->xgen;
    }
  }
  HIPCHECK(hipGetLastError());
  {
    int32_t flag = 0;
    HIPCHECK(hipMemcpyAsync(&flag, A + o_flag, sizeof(flag), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (flag) return fail(NMFK_ERR_NAN_INIT, "Initial values for the W/H matrix entries include NaNs!");
  }

  NmfkStepArgs hs;
  hs.arena = A;
  hs.X = ctx->Xr;
  hs.Xalt = ctx->Xc;
  hs.Xtile = any_hyb ? (const float *)ctx->xtile.p : nullptr;
  hs.ld = m;
  hs.L = m;
  hs.D = n;
  hs.S = ghp[0].S;
  hs.dchunk = ghp[0].dchunk;
  hs.wsplit = ghp[0].wsplit;
  hs.fused = ghp[0].fused;
  hs.PW = PW;
  hs.PH = PH;
  hs.which = 0;
  hs.it = 0;
  hs.has_nan = ctx->nan_count > 0;
  hs.lambda = (float)ctx->lambda;
  hs.runs = d_runs;
  hs.state = d_state;
  hs.nunits = nunits;
  hs.force = 0;
  hs.res_wgs = res_wgs[0];
  hs.clampw = 0;
  hs.fuse_red = 0;
  hs.lag = T.hyb_lag;
  hs.bnum = T.wide_bn;
  NmfkStepArgs ws = hs;
  ws.res_wgs = res_wgs[1];
  ws.X = ctx->Xc;
  ws.Xalt = ctx->Xr;
  ws.Xtile = any_hyb ? (const float *)ctx->xtile.p + tile_h : nullptr;
  ws.ld = n;
  ws.L = n;
  ws.D = m;
  ws.S = gwp[0].S;
  ws.dchunk = gwp[0].dchunk;
  ws.wsplit = gwp[0].wsplit;
  ws.fused = gwp[0].fused;
  ws.which = 1;
  // one pair of half-step argument blocks per phase (they differ in the launch geometry only)
  NmfkStepArgs hsP[2] = {hs, hs}, wsP[2] = {ws, ws};
  hsP[1].S = ghp[1].S;
  hsP[1].dchunk = ghp[1].dchunk;
  hsP[1].wsplit = ghp[1].wsplit;
  hsP[1].fused = ghp[1].fused;
  wsP[1].S = gwp[1].S;
  wsP[1].dchunk = gwp[1].dchunk;
  wsP[1].wsplit = gwp[1].wsplit;
  wsP[1].fused = gwp[1].fused;

  // Deferred check (see the loop): is the sweep eligible, and does a pair of half-step geometries allow it -- the H half-step
  // must have an objective mode for its geometry, and the W half-step must end on a kernel that clamps what it writes in a check
  // iteration (NmfkStepArgs::clampw)
  const bool defer_any = T.defer_obj && !f64 && !P.Hfixed && !P.Wfixed && ctx->Wgt == nullptr &&
                         P.weight > 0;  // (the launchers take the weight as the switch: objw > 0)
  const bool defer_ok = defer_any && !ctx->sparse && T.hyb_sse && T.wide_sse;
  // sparse X, blocked form (ranks up to 32): the H half-step's products at the non-zeros are the objective's; the Gram term keeps its
  // launches.  Needs both half-steps in the blocked form and a slot per lane tile of the H half-step behind slot 0 of ossepart.
  const bool defer_sp = defer_any && ctx->sparse && sp_blk[0] && sp_blk[1] && (m + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS <= tiles_n;
  auto defer_geo = [&](const NmfkStepArgs &h, const NmfkStepArgs &w, bool hyb) {
    NmfkStepArgs hh = h;
    if (!hyb) hh.res_wgs = 0;  // (the resident form is the rank <= 16 kernels' only)
    if (!hyb && hh.wsplit > 1) return 0;        // (the wide-rank kernel has no form for per-wave loop ranges)
    const int parts = nmfk_hyb_step_parts(hh);  // (and the same lane tile)
    const bool wfin = hyb || !w.fused || w.wsplit == 1;  // the W half-step ends on a kernel that clamps (the matrix-pipe kernels' fused finishes, reduce_kernel)
    return (defer_ok && parts > 0 && parts <= obj_cap && wfin) ? parts : 0;
  };
  // groups whose kernels have the objective mode: the rank <= 16 matrix-pipe group(s) and the split-operand wide-rank kernel
  auto defer_kind = [&](const Group &G) {
    if (ctx->sparse) return (defer_sp && nmfk_sp_blk_rank(G.kp)) ? 3 : 0;
    return G.hyb != 0 ? 1 : (use_wide_k(G.k) && use_wide2_k(G.k) && G.kp != 0) ? 2 : 0;
  };
  for (int ph = 0; ph < 2; ++ph) {
    bool any = false, all = true;
    for (const Group &G : groups)
      if (G.phase == ph && defer_kind(G)) {
        any = true;
        all = all && defer_geo(hsP[ph], wsP[ph], defer_kind(G) == 1) > 0;
      }
    wsP[ph].clampw = (any && all) ? std::max(1, (int)P.maxiter) : 0;  // (the value is maxiter: not in the last iteration's classic check)
  }
  // Fused reduce (round 5): the H half-step of a matrix-pipe group whose loop range is split over workgroups (S > 1: few units) leaves
  // partial numerators; when the W half-step behind it runs the resident form, that launch sums them while it stages H
  // (NmfkStepArgs::fuse_red) and the reduce launch between the two half-steps is not queued.  Needs both half-steps to run every
  // iteration and H's finish to be the plain one.
  auto fuse_red_of = [&](const NmfkStepArgs &h, const NmfkStepArgs &w) {
    return (T.fuse_red && !P.Hfixed && !P.Wfixed && !h.fused && h.res_wgs == 0 && h.S > 1 && w.res_wgs > 0) ? h.S : 0;
  };
  for (int ph = 0; ph < 2; ++ph) {
    bool all_hyb_ph = true, any_ph = false;  // (the phase's argument block is shared by its groups: all of them on these kernels)
    for (const Group &G : groups)
      if (G.phase == ph) {
        any_ph = true;
        all_hyb_ph = all_hyb_ph && G.hyb != 0;
      }
    wsP[ph].fuse_red = (any_ph && all_hyb_ph) ? fuse_red_of(hsP[ph], wsP[ph]) : 0;
    hsP[ph].fuse_red = wsP[ph].fuse_red;  // (the H half-step leaves the W half-step its copy of colsum(W): NmfkRun::osnapW)
  }
  // device copies of the half-step argument blocks (constant over the sweep; `it` is passed by value)
  const NmfkStepArgs *d_hsP[2] = {(const NmfkStepArgs *)(A + o_args), (const NmfkStepArgs *)(A + o_args) + 2};
  const NmfkStepArgs *d_wsP[2] = {d_hsP[0] + 1, d_hsP[1] + 1};
  {
    NmfkStepArgs all4[4] = {hsP[0], wsP[0], hsP[1], wsP[1]};
    HIPCHECK(hipMemcpy(A + o_args, all4, sizeof(all4), hipMemcpyHostToDevice));
  }

  NmfkSseArgs sa;
  sa.arena = A;
  sa.Xc = ctx->Xc;
  sa.Xr = ctx->Xr;
  sa.Wgt = ctx->Wgt;
  sa.n = n;
  sa.m = m;
  sa.hsel = 0;
  sa.weight = P.weight;
  sa.runs = d_runs;
  sa.state = d_state;
  sa.nunits = nunits;
  sa.force = 0;
  sa.total_iters = 0;

  NmfkCheckArgs ca;
  ca.arena = A;
  ca.n = n;
  ca.m = m;
  ca.it = 0;
  ca.ntile_n = ctx->sparse ? tiles_n + 1 : tiles_n;
  ca.PW = PW;
  ca.PH = PH;
  ca.tol = P.tol;
  ca.tolOF = P.tolOF;
  ca.maxiter = P.maxiter;
  ca.maxbaditers = P.maxbaditers;
  ca.maxreattempts = P.maxreattempts;
  ca.stopconv = P.stopconv;
  ca.runs = d_runs;
  ca.state = d_state;
  ca.nunits = nunits;
  ca.trace = trace_stride ? (double *)(A + o_trace) : nullptr;
  ca.trace_stride = trace_stride;
  ca.track_low = 0;
  ca.w_clamped = 0;

  NmfkSparseArgs sph, spw;  // CSC view (H half-step), CSR view (W half-step, objective)
  sph.arena = A;
  sph.ptr = ctx->colptr;
  sph.rec = ctx->rec_csc;
  sph.nrec = (int32_t)std::max<int64_t>(ctx->nnz, 1);
  sph.runs = d_runs;
  sph.state = d_state;
  sph.L = m;
  sph.which = 0;
  sph.it = 0;
  sph.PW = PW;
  sph.PH = PH;
  sph.force = 0;
  sph.split = 1;
  sph.ell = sp_blk[0] ? ctx->ell[1] : nullptr;
  sph.ellptr = ctx->ellptr[1];
  sph.ngb = ctx->ell_ngb[1];
  sph.D = n;
  sph.objw = 0.0;
  sph.ntile_obj = tiles_n;
  sph.clampw = 0;
  spw = sph;
  spw.split = 0;
  spw.ell = sp_blk[1] ? ctx->ell[0] : nullptr;
  spw.ellptr = ctx->ellptr[0];
  spw.ngb = ctx->ell_ngb[0];
  spw.D = m;
  spw.ptr = ctx->rowptr;
  spw.rec = ctx->rec_csr;
  spw.L = n;
  spw.which = 1;
  const bool sparse = ctx->sparse;

  // Rank groups run concurrently: group j owns stream j mod NS.  Inside a group the order H half-step ->
  // W half-step -> (every 10th iteration) objective + check block is the stream order; different ranks never
  // exchange data before the clustering step, so no cross-stream synchronisation is needed inside the loop.
  const int ngroups = (int)groups.size();
  auto use_wide = [&](const Group &G) { return use_wide_k(G.k); };
  const bool wide_sse = T.wide_sse != 0;
  auto use_hyb = [&](const Group &G) { return G.hyb != 0; };
  const bool hyb_sse = T.hyb_sse != 0;  // objective of those groups on the matrix pipe
  // sparse X: the (few, large) launch groups run one after the other -- side by side their gathers evict each other's
  // factor rows from L2 (29.2 vs 31.7 ms per iteration on BASELINE configs[3])
  // dense X: per-rank launch groups (ranks above 16, fp64 compute) side by side on up to 8 streams, 16 when there are a dozen
  // groups or more on a matrix that is not tiny (k = 17:32 x 8 at 8192 x 512: 1.37 -> 1.05 ms per iteration, at 20000 x 1000 5.9 -> 5.0;
  // one after the other: 6.3 ms; at 1024 x 128 the loop is bound by the host's launches and more streams buy nothing --
  // profiles/r04/wide_rank_streams.txt)
  const int max_streams = T.streams > 0 ? T.streams : ctx->sparse ? 1 : (ngroups >= 12 && (double)n * m >= 0.25 * 8192.0 * 512.0) ? 16 : 8;
  int ncohorts = 0;  // stream slots: one per cohort of every launch group
  std::vector<std::vector<int>> coh_stream(groups.size());
  for (size_t j = 0; j < groups.size(); ++j) {
    size_t parts = coh[j].size();
    if (replanning && j == 0)  // (a later tier may deal the units still active to more cohorts than the sweep starts with)
      for (const Tier &t : tiers) parts = std::max(parts, (size_t)t.ncoh);
    for (size_t c = 0; c < parts; ++c) coh_stream[j].push_back(ncohorts++);
  }
  const int NS = std::min(ncohorts, max_streams);
  for (auto &v : coh_stream)
    for (int &x : v) x %= NS;
  while ((int)ctx->gstreams.size() < NS) {
    hipStream_t gs;
    HIPCHECK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
    ctx->gstreams.push_back(gs);
  }
  if (!ctx->poll_stream) HIPCHECK(hipStreamCreateWithFlags(&ctx->poll_stream, hipStreamNonBlocking));
  hipStream_t poll = ctx->poll_stream;
  // the sweep's events live in a holder that destroys them on EVERY way out (the error returns of HIPCHECK / fail()
  // below included)
  struct EventBag {
    std::vector<hipEvent_t> all;
    ~EventBag() {
      for (hipEvent_t e : all) (void)hipEventDestroy(e);
    }
    hipError_t make(hipEvent_t *e, unsigned flags) {
      const hipError_t r = hipEventCreateWithFlags(e, flags);
      if (r == hipSuccess) all.push_back(*e);
      return r;
    }
  } events;
  std::vector<hipEvent_t> gev(2 * (size_t)NS);
  for (auto &e : gev) HIPCHECK(events.make(&e, hipEventDisableTiming));
  std::vector<hipEvent_t> rp_ev((size_t)NS + 1);  // re-plans: join / fork of the cohort streams
  for (auto &e : rp_ev) HIPCHECK(events.make(&e, hipEventDisableTiming));
  hipEvent_t snap_ev[2], start_ev;
  HIPCHECK(events.make(&snap_ev[0], hipEventDisableTiming));
  HIPCHECK(events.make(&snap_ev[1], hipEventDisableTiming));
  HIPCHECK(events.make(&start_ev, hipEventDisableTiming));
  hipEvent_t loop_t0 = nullptr, loop_t1 = nullptr;
  if (ctx->profiling) {
    HIPCHECK(events.make(&loop_t0, hipEventDefault));
    HIPCHECK(events.make(&loop_t1, hipEventDefault));
    HIPCHECK(hipEventRecord(loop_t0, st));
  }
  HIPCHECK(hipEventRecord(start_ev, st));  // init done
  for (int j = 0; j < NS; ++j) HIPCHECK(hipStreamWaitEvent(ctx->gstreams[j], start_ev, 0));

  // Mult:64 guard before the first iteration
  const bool guard0 = P.maxiter > 0 && P.maxbaditers > 0 && P.maxreattempts > 0;
  NmfkState *snap[2] = {(NmfkState *)ctx->pinned, (NmfkState *)ctx->pinned + nunits};
  int total_iters = 0;
  const int maxiter = guard0 ? (int)P.maxiter : 0;
  double host_wait_s = 0;  // time the host spent waiting for the GPU inside the loop (NMFK_HOST_TIMING=1 prints it)
  const auto loop_w0 = std::chrono::steady_clock::now();
  int nphases = 1;
  for (const Group &G : groups) nphases = std::max(nphases, G.phase + 1);
  {
    int32_t *si = ctx->sweep_info;
    memset(si, 0, sizeof(ctx->sweep_info));
    si[0] = nphases;
    si[3] = ngroups;
    for (const Group &G : groups) si[2] += (G.kp == 0 && !G.hyb) ? 1 : 0;
    for (int u = 0; u < nunits; ++u) {
      si[1] += runs[u].hyb ? 1 : 0;
      si[4] += use_wide_k(runs[u].k) ? 1 : 0;
    }
  }
  std::vector<char> in_phase(nunits);
  // retire-aware schedule: state of the re-plans (see the tiers above)
  const std::vector<Group> groups0 = groups;  // (the profile below counts over the launch groups as they started)
  const std::vector<NmfkRun> runs0 = runs;    // (units by NmfkRun::uid = position in the list the sweep starts with)
  int cur_tier = 0, cur_buf = 0, nreplans = 0, snap_epoch[2] = {0, 0};
  std::vector<std::vector<int32_t>> order_hist(1, std::vector<int32_t>((size_t)nunits));  // [re-plan][position] = NmfkRun::uid
  for (int u = 0; u < nunits; ++u) order_hist[0][(size_t)u] = u;
  std::deque<std::vector<int32_t>> perm_keep;             // host sources of the asynchronous uploads
  std::deque<std::array<NmfkStepArgs, 2>> args_keep;
  // (the tiers were planned for ONE launch group; should the group builder ever disagree, the sweep keeps its first plan -- a pure optimisation must
  //  not fail a user's execute(): ADVICE r4 / VERDICT r5.  The buffers laid out for the re-plans stay unused.)
  bool replan_live = replanning;
  if (replanning && ngroups != 1) {
    if (T.debug) fprintf(stderr, "[nmfk] the retire-aware schedule expects one launch group, the sweep has %d: static schedule\n", ngroups);
    replan_live = false;
  }
  // Deferred check (round 4).  The H half-step of iteration j + 1 forms W*H of exactly the factors whose objective the check
  // after iteration j monitors (Mult:74), so on the matrix-pipe kernels that half-step leaves the objective as a by-product and
  // the launch that recomputed W*H for it (0.39 of the check block's 0.52 ms on the bench sweep) goes away: the check iteration
  // runs only the clamp (Mult:99-100), the tolerance / stagnation / consistency tests follow the next H half-step.  H is
  // double-buffered by iteration parity and that half-step writes nothing else a stopped unit keeps, so a unit the check
  // retires ends with the factors of iteration j as before (NmfkState::iters = j selects the buffer).  Two differences from
  // the plain order, both far below fp32 rounding: the objective is that of the factors AFTER the clamp (entries below eps()
  // raised to eps(): <= 1e-13 of the objective), and a unit that stops on `tol` (Mult:75-78) keeps clamped factors.
  // Not for the last iteration (no half-step follows), fixed factors, array weights -- those checks keep their objective launch.
  auto defer_parts = [&](const Group &G) {
    if (defer_kind(G) == 3) return tiles_n + 1;  // (what check_a adds: slot 0 = the Gram term, the H half-step's lane tiles, zeros)
    return (defer_kind(G) && wsP[G.phase].clampw) ? defer_geo(hsP[G.phase], wsP[G.phase], defer_kind(G) == 1) : 0;
  };
  auto track_low_of = [&](const Group &G) {
    const NmfkStepArgs &hs = hsP[G.phase], &ws = wsP[G.phase];
    return (int)(!T.clamp_always && use_hyb(G) && !P.Hfixed && !P.Wfixed && (hs.fused || hs.res_wgs > 0) && (ws.fused || ws.res_wgs > 0));
  };
  int max_cohorts = 1;
  for (const auto &v : coh) max_cohorts = std::max(max_cohorts, (int)v.size());
  std::vector<int> pending((size_t)ngroups, 0);  // > 0: the group's check of the previous iteration waits for this H half-step (= its objective partials per unit)
  int ndeferred = 0, nclassic = 0, nfused_red = 0;
  for (int phase = 0; phase < nphases; ++phase) {
  // (units still active when a phase's loop ends ran all `maxiter` iterations, so one total_iters serves every phase)
  for (int u = 0; u < nunits; ++u) in_phase[u] = 0;
  for (const Group &G : groups)
    if (G.phase == phase)
      for (int u = G.begin; u < G.begin + G.count; ++u) in_phase[u] = 1;
  int nchecks = 0;
  bool all_done = !guard0;
  std::fill(pending.begin(), pending.end(), 0);
  for (int it = 0; it < maxiter && !all_done; ++it) {
    const bool check = (it + 1) % 10 == 0;  // Mult:73
    const bool timed = prof.want(it);
    bool completing = false, deferring = false;  // deferred checks end / start in this iteration
    hsP[0].it = hsP[1].it = wsP[0].it = wsP[1].it = it;
    sa.hsel = (it + 1) & 1;
    ca.it = it;
    for (int j = 0; j < ngroups; ++j) {
      const Group &G = groups[j];
      if (G.phase != phase) continue;
      const NmfkStepArgs &hs = hsP[G.phase], &ws = wsP[G.phase];
      const NmfkStepArgs *d_hs = d_hsP[G.phase], *d_ws = d_wsP[G.phase];
      const int pend = pending[j];  // (the same for every cohort of the group)
      const int epoch = (int)order_hist.size() - 1;
      for (size_t cq = 0; cq < coh[j].size(); ++cq) {
      const int ub = coh[j][cq].first, uc = coh[j][cq].second;
      if (uc <= 0) continue;
      hipStream_t gs = ctx->gstreams[coh_stream[j][cq]];
      if (!P.Hfixed) {  // Mult:66-68
        const size_t e0 = timed ? prof.begin(gs) : 0;
        sph.it = it;
        sph.objw = (sparse && pend) ? P.weight : 0.0;
        if (sparse && f64)
          nmfk_launch_sp_step_f64(&sph, G.kp, ub, uc, gs);
        else if (sparse)
          nmfk_launch_sp_step_f32(&sph, G.kp, ub, uc, gs);
        else if (use_hyb(G))
          nmfk_launch_step_hyb_f32(hs, d_hs, G.hyb, ub, uc, gs, pend ? P.weight : 0.0);
        else if (G.kp == 0 && f64)
          nmfk_launch_step_multi_f64(hs, d_hs, ub, uc, gs);
#if NMFK_WITH_MERGED_F32
        else if (G.kp == 0)
          nmfk_launch_step_multi_f32(hs, d_hs, ub, uc, gs);
#endif
        else if (f64)
          nmfk_launch_step_f64(hs, d_hs, G.kp, ub, uc, gs);
        else if (use_wide(G) && use_wide2_k(G.k) && hs.wsplit == 1)
          nmfk_launch_step_wide2_f32(hs, d_hs, G.kp, ub, uc, gs, pend ? P.weight : 0.0);
        else if (use_wide(G))
          nmfk_launch_step_mfma_wide_f32(hs, d_hs, G.kp, ub, uc, gs);
        else
          nmfk_launch_step_f32(hs, d_hs, G.kp, ub, uc, gs);
        if (timed) prof.end(e0, PK_HSTEP, j, it, gs, ub, uc, epoch);
        if (!hs.fused && use_hyb(G) && hs.res_wgs == 0 && ws.fuse_red > 0) ++nfused_red;
        if (!hs.fused && !(use_hyb(G) && hs.res_wgs > 0) && !(use_hyb(G) && ws.fuse_red > 0)) {  // (the resident form always finishes
          if (f64)                                                                                  //  itself; fuse_red: the W half-step does)
            nmfk_launch_reduce_f64(hs, ub, uc, gs);
          else
            nmfk_launch_reduce_f32(hs, ub, uc, gs);
        }
        if (pend) {  // the deferred check of iteration it - 1: the objective has just been left by the half-step
          NmfkCheckArgs cb = ca;
          cb.it = it - 1;
          cb.ntile_n = pend;  // (the partials this launch has left: defer_parts at the check iteration)
          cb.track_low = track_low_of(G);
          nmfk_launch_check_f32(cb, ub, uc, gs, 1 | 4);
          completing = true;
        }
      }
      if (!P.Wfixed) {  // Mult:69-71
        const size_t e0 = timed ? prof.begin(gs) : 0;
        spw.it = it;
        spw.clampw = defer_kind(G) == 3 ? maxiter : 0;
        if (sparse && f64)
          nmfk_launch_sp_step_f64(&spw, G.kp, ub, uc, gs);
        else if (sparse)
          nmfk_launch_sp_step_f32(&spw, G.kp, ub, uc, gs);
        else if (use_hyb(G))
          nmfk_launch_step_hyb_f32(ws, d_ws, G.hyb, ub, uc, gs);
        else if (G.kp == 0 && f64)
          nmfk_launch_step_multi_f64(ws, d_ws, ub, uc, gs);
#if NMFK_WITH_MERGED_F32
        else if (G.kp == 0)
          nmfk_launch_step_multi_f32(ws, d_ws, ub, uc, gs);
#endif
        else if (f64)
          nmfk_launch_step_f64(ws, d_ws, G.kp, ub, uc, gs);
        else if (use_wide(G) && use_wide2_k(G.k) && ws.wsplit == 1)
          nmfk_launch_step_wide2_f32(ws, d_ws, G.kp, ub, uc, gs);
        else if (use_wide(G))
          nmfk_launch_step_mfma_wide_f32(ws, d_ws, G.kp, ub, uc, gs);
        else
          nmfk_launch_step_f32(ws, d_ws, G.kp, ub, uc, gs);
        if (timed) prof.end(e0, PK_WSTEP, j, it, gs, ub, uc, epoch);
        if (!ws.fused && !(use_hyb(G) && ws.res_wgs > 0)) {
          NmfkStepArgs wr = ws;  // (the phase's argument block serves groups on other kernels too: only this group's choice counts)
          wr.clampw = defer_kind(G) != 0 ? ws.clampw : 0;
          if (f64)
            nmfk_launch_reduce_f64(wr, ub, uc, gs);
          else
            nmfk_launch_reduce_f32(wr, ub, uc, gs);
        }
      }
      ca.w_clamped = it + 1 < maxiter && (defer_kind(G) == 3 || (defer_kind(G) != 0 && wsP[G.phase].clampw));
      if (check && it + 1 < maxiter && defer_parts(G) > 0) {
        ca.track_low = track_low_of(G);
        nmfk_launch_check_f32(ca, ub, uc, gs, 2);
        if (sparse) {  // the Gram term of the clamped factors -> slot 0; the non-zero terms come with the next H half-step
          spw.it = it;
          nmfk_launch_sp_obj_f32(&spw, n, m, (it + 1) & 1, 0, P.weight, ub, uc, gs, 2);
        }
        deferring = true;
      } else if (check) {
        if (sparse) {
          spw.it = it;
          if (f64)
            nmfk_launch_sp_obj_f64(&spw, n, m, (it + 1) & 1, 0, P.weight, ub, uc, gs);
          else
            nmfk_launch_sp_obj_f32(&spw, n, m, (it + 1) & 1, 0, P.weight, ub, uc, gs);
        } else if (f64) {
          nmfk_launch_sse_f64(sa, ub, uc, gs);
        } else if (use_hyb(G) && sa.Wgt == nullptr && hyb_sse) {
          nmfk_launch_hyb_sse(wsP[G.phase], d_wsP[G.phase], P.weight, (it + 1) & 1, G.hyb, ub, uc, gs);
        } else if (use_wide(G) && use_wide2_k(G.k) && sa.Wgt == nullptr && wide_sse) {
          nmfk_launch_wide2_sse(wsP[G.phase], d_wsP[G.phase], P.weight, (it + 1) & 1, G.kp, ub, uc, gs);
        } else if (use_wide(G) && sa.Wgt == nullptr && wide_sse) {
          nmfk_launch_sse_mfma_wide_f32(sa, G.kp, ub, uc, gs);
        } else {
          nmfk_launch_sse_f32(sa, ub, uc, gs);
        }
        // the clamp pass only where a value below eps() may exist: the matrix-pipe kernels' fused finishes watch what they write
        // in a check iteration (both half-steps run and finish themselves; NmfkState::lowflag)
        ca.track_low = track_low_of(G);
        if (f64)
          nmfk_launch_check_f64(ca, ub, uc, gs);
        else
          nmfk_launch_check_f32(ca, ub, uc, gs);
      }
      }  // cohorts
      if (pend && !P.Hfixed) pending[j] = 0;
      if (check && it + 1 < maxiter && defer_parts(G) > 0) {
        pending[j] = defer_parts(G);
        ++ndeferred;
      } else if (check) {
        ++nclassic;
      }
    }
    total_iters = std::max(total_iters, it + 1);
    if ((check && !deferring) || completing) {  // every group's check of this cadence is queued: snapshot of the states, re-plan
      const int slot = nchecks & 1;
      int next_tier = cur_tier, act = 0;
      if (nchecks > 0) {  // inspect the PREVIOUS check while this one is still queued
        const auto w0 = std::chrono::steady_clock::now();
        HIPCHECK(hipEventSynchronize(snap_ev[slot ^ 1]));
        host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        bool any = false;
        for (int u = 0; u < nunits; ++u) any = any || (in_phase[u] && snap[slot ^ 1][u].active);
        if (!any) all_done = true;
        if (replan_live && any) {  // units still active as of that check: do they fit a later tier?
          for (int u = 0; u < nunits; ++u) act += snap[slot ^ 1][u].active ? 1 : 0;
          while (next_tier + 1 < (int)tiers.size() && act <= tiers[next_tier + 1].count) ++next_tier;
        }
      }
      for (int j = 0; j < NS; ++j) {
        HIPCHECK(hipEventRecord(gev[slot * NS + j], ctx->gstreams[j]));
        HIPCHECK(hipStreamWaitEvent(poll, gev[slot * NS + j], 0));
      }
      HIPCHECK(hipMemcpyAsync(snap[slot], d_state, sizeof(NmfkState) * nunits, hipMemcpyDeviceToHost, poll));
      HIPCHECK(hipEventRecord(snap_ev[slot], poll));
      snap_epoch[slot] = (int)order_hist.size() - 1;
      nchecks++;
      if (next_tier != cur_tier && it + 1 < maxiter) {
        // ---- re-plan (takes effect with the next iteration; everything below is queued behind this check on the group's
        // stream, the host does not wait).  The snapshot just inspected may predate an earlier re-plan: unit ids translate.
        hipStream_t gs = ctx->gstreams[coh_stream[0][0]];
        // (cohorts: the work list is permuted as a whole, so the re-plan waits for every cohort's stream and they for it)
        for (int j = 0; j < NS; ++j)
          if (ctx->gstreams[j] != gs) {
            HIPCHECK(hipEventRecord(rp_ev[j], ctx->gstreams[j]));
            HIPCHECK(hipStreamWaitEvent(gs, rp_ev[j], 0));
          }
        const std::vector<int32_t> &seen = order_hist[(size_t)snap_epoch[slot ^ 1]], &cur = order_hist.back();
        std::vector<char> alive((size_t)nunits, 0);
        for (int u = 0; u < nunits; ++u) alive[(size_t)seen[u]] = snap[slot ^ 1][u].active ? 1 : 0;
        perm_keep.emplace_back((size_t)nunits);
        std::vector<int32_t> &perm = perm_keep.back(), order((size_t)nunits);
        const Tier &tr = tiers[(size_t)next_tier];
        int w = 0;
        {  // the units still active first -- dealt round-robin, in their present order, to the tier's cohorts (every cohort the same
          // mix of ranks) --, then the others
          std::vector<int> act_pos;
          for (int u = 0; u < nunits; ++u)
            if (alive[(size_t)cur[u]]) act_pos.push_back(u);
          const int na = (int)act_pos.size(), nc = std::max(1, std::min(tr.ncoh, na));
          for (int c = 0; c < nc; ++c)
            for (int i = c; i < na; i += nc) {
              perm[(size_t)w] = act_pos[(size_t)i];
              order[(size_t)w++] = cur[(size_t)act_pos[(size_t)i]];
            }
          for (int u = 0; u < nunits; ++u)
            if (!alive[(size_t)cur[u]]) {
              perm[(size_t)w] = u;
              order[(size_t)w++] = cur[u];
            }
        }
        int32_t *d_perm = (int32_t *)(A + o_perm) + (size_t)nreplans * nunits;
        HIPCHECK(hipMemcpyAsync(d_perm, perm.data(), sizeof(int32_t) * nunits, hipMemcpyHostToDevice, gs));
        hipLaunchKernelGGL(replan_kernel, dim3(nunits), dim3(64), 0, gs, A, d_runs_buf[cur_buf], d_state_buf[cur_buf],
                           d_runs_buf[cur_buf ^ 1], d_state_buf[cur_buf ^ 1], d_perm, tr.nsW, tr.nsH, tiers[(size_t)cur_tier].PW,
                           tiers[(size_t)cur_tier].PH, std::max(tiers[(size_t)cur_tier].PW, tr.PW), std::max(tiers[(size_t)cur_tier].PH, tr.PH),
                           (int64_t)o_partpool, (int64_t)part_stride[(size_t)next_tier], tr.count);
        cur_buf ^= 1;
        d_runs = d_runs_buf[cur_buf];
        d_state = d_state_buf[cur_buf];
        {
          std::vector<NmfkRun> moved((size_t)nunits);
          for (int u = 0; u < nunits; ++u) {
            moved[(size_t)u] = runs[(size_t)perm[(size_t)u]];
            moved[(size_t)u].nsW = tr.nsW;
            moved[(size_t)u].nsH = tr.nsH;
            moved[(size_t)u].opart = (int64_t)(o_partpool + (size_t)(u < tr.count ? u : 0) * part_stride[(size_t)next_tier]);
          }
          runs.swap(moved);
        }
        for (int ph = 0; ph < 2; ++ph) {
          NmfkStepArgs *two[2] = {&hsP[ph], &wsP[ph]};
          for (int f = 0; f < 2; ++f) {
            const Geo &g = f == 0 ? tr.gh : tr.gw;
            two[f]->S = g.S;
            two[f]->dchunk = g.dchunk;
            two[f]->wsplit = g.wsplit;
            two[f]->fused = g.fused;
            two[f]->res_wgs = tr.res[f];
            two[f]->PW = tr.PW;
            two[f]->PH = tr.PH;
            two[f]->runs = d_runs;
            two[f]->state = d_state;
          }
          wsP[ph].clampw = defer_geo(hsP[ph], wsP[ph], true) > 0 ? std::max(1, (int)P.maxiter) : 0;
          wsP[ph].fuse_red = fuse_red_of(hsP[ph], wsP[ph]);
          hsP[ph].fuse_red = wsP[ph].fuse_red;
        }
        args_keep.push_back({hsP[0], wsP[0]});
        NmfkStepArgs *d_two = (NmfkStepArgs *)(A + o_args2) + 2 * (size_t)nreplans;
        HIPCHECK(hipMemcpyAsync(d_two, args_keep.back().data(), 2 * sizeof(NmfkStepArgs), hipMemcpyHostToDevice, gs));
        d_hsP[0] = d_hsP[1] = d_two;
        d_wsP[0] = d_wsP[1] = d_two + 1;
        sa.runs = ca.runs = d_runs;
        sa.state = ca.state = d_state;
        ca.PW = tr.PW;
        ca.PH = tr.PH;
        groups[0].count = act;
        {  // the units still active are dealt out again: equal contiguous parts (the list is rank-sorted inside each old cohort)
          const int nc = std::max(1, std::min(tr.ncoh, act));
          coh[0].resize((size_t)nc);
          max_cohorts = std::max(max_cohorts, nc);
          int b = groups[0].begin;
          for (int c = 0; c < nc; ++c) {
            const int cnt_c = act / nc + (c < act % nc ? 1 : 0);
            coh[0][(size_t)c] = {b, cnt_c};
            b += cnt_c;
          }
        }
        HIPCHECK(hipEventRecord(rp_ev[NS], gs));
        for (int j = 0; j < NS; ++j)
          if (ctx->gstreams[j] != gs) HIPCHECK(hipStreamWaitEvent(ctx->gstreams[j], rp_ev[NS], 0));
        order_hist.push_back(order);
        cur_tier = next_tier;
        ++nreplans;
      }
    }
  }
  if (phase + 1 < nphases) {  // the next phase starts on an empty GPU
    for (int j = 0; j < NS; ++j) HIPCHECK(hipStreamSynchronize(ctx->gstreams[j]));
    HIPCHECK(hipStreamSynchronize(poll));
  }
  }  // phases
  ctx->sweep_info[5] = nreplans;
  ctx->sweep_info[6] = cur_tier;
  ctx->sweep_info[7] = ngroups == 1 ? groups[0].count : 0;
  ctx->sweep_info[10] = max_cohorts;  // (the most the sweep ran side by side; a late tier of a handful of units runs as one)
  ctx->sweep_info[11] = nfused_red;
  ctx->sweep_info[8] = ndeferred;
  ctx->sweep_info[9] = nclassic;
  if (T.host_timing)
    fprintf(stderr, "[nmfk] loop: %d iterations, %d groups, host %.3f s of which waiting for the GPU %.3f s\n", total_iters,
            ngroups, std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_w0).count(), host_wait_s);
  HIPCHECK(hipGetLastError());
  // join: the main stream continues after every group stream (and the poll stream) has drained
  for (int j = 0; j < NS; ++j) {
    HIPCHECK(hipEventRecord(gev[j], ctx->gstreams[j]));
    HIPCHECK(hipStreamWaitEvent(st, gev[j], 0));
  }
  HIPCHECK(hipStreamSynchronize(poll));
  if (ctx->profiling) HIPCHECK(hipEventRecord(loop_t1, st));

  // objvalue = normnan(X - W*H) on the final factors, normalisation, T-typed outputs (Exec:790-805)
  sa.hsel = -1;
  sa.force = 1;
  sa.total_iters = total_iters;
  // Mult:125: sum(((X - W*H) .* weight)[.!inan].^2) on the final factors (only needed when a weight is in play)
  const int fa_ntile = ctx->sparse ? tiles_n + 1 : tiles_n;
  const size_t o_sse = 0;
  const bool weighted = ctx->Wgt != nullptr || P.weight != 1.0;
  spw.force = 1;
  if (weighted && sse_out) {
    if (ctx->scratch.ensure(sizeof(double) * (size_t)nunits)) return fail(NMFK_ERR_HIP, "out of device memory");
    if (sparse && f64)
      nmfk_launch_sp_obj_f64(&spw, n, m, -1, total_iters, P.weight, 0, nunits, st);
    else if (sparse)
      nmfk_launch_sp_obj_f32(&spw, n, m, -1, total_iters, P.weight, 0, nunits, st);
    else if (f64)
      nmfk_launch_sse_f64(sa, 0, nunits, st);
    else
      nmfk_launch_sse_f32(sa, 0, nunits, st);
    if (f64)
      nmfk_launch_sum_parts_f64(A, d_runs, nunits, fa_ntile, (double *)(ctx->scratch.p + o_sse), st);
    else
      nmfk_launch_sum_parts_f32(A, d_runs, nunits, fa_ntile, (double *)(ctx->scratch.p + o_sse), st);
  }
  sa.weight = 1.0;
  sa.Wgt = nullptr;
  NmfkFinishArgs fa;
  fa.arena = A;
  fa.n = n;
  fa.m = m;
  fa.ntile_n = ctx->sparse ? tiles_n + 1 : tiles_n;
  fa.total_iters = total_iters;
  fa.normalize = P.normalize;
  fa.runs = d_runs;
  fa.state = d_state;
  fa.nunits = nunits;
  fa.Wout = (float *const *)(d_ptrs + 2 * nk);
  fa.Hout = (float *const *)(d_ptrs + 3 * nk);
  fa.frob = (float *const *)(d_ptrs + 4 * nk);
  fa.iters = (int32_t *const *)(d_ptrs + 5 * nk);
  fa.reason = (int32_t *const *)(d_ptrs + 6 * nk);
  if (sparse && f64)
    nmfk_launch_sp_obj_f64(&spw, n, m, -1, total_iters, 1.0, 0, nunits, st);
  else if (sparse)
    nmfk_launch_sp_obj_f32(&spw, n, m, -1, total_iters, 1.0, 0, nunits, st);
  else if (f64)
    nmfk_launch_sse_f64(sa, 0, nunits, st);
  else
    nmfk_launch_sse_f32(sa, 0, nunits, st);
  if (f64)
    nmfk_launch_finish_f64(fa, st);
  else
    nmfk_launch_finish_f32(fa, st);
  HIPCHECK(hipGetLastError());

  std::vector<std::vector<float>> h_frob(nk);
  std::vector<std::vector<int32_t>> h_iters(nk);
  for (int q = 0; q < nk; ++q) {
    const size_t k = (size_t)ks[q];
    HIPCHECK(hipMemcpyAsync(W_out[q], A + o_Wo[q], sizeof(float) * nruns * k * n, hipMemcpyDefault, st));
    HIPCHECK(hipMemcpyAsync(H_out[q], A + o_Ho[q], sizeof(float) * nruns * k * m, hipMemcpyDefault, st));
    HIPCHECK(hipMemcpyAsync(frob_out[q], A + o_frob[q], sizeof(float) * nruns, hipMemcpyDefault, st));
    if (iters_out && iters_out[q])
      HIPCHECK(hipMemcpyAsync(iters_out[q], A + o_iters[q], sizeof(int32_t) * nruns, hipMemcpyDefault, st));
    if (reason_out && reason_out[q])
      HIPCHECK(hipMemcpyAsync(reason_out[q], A + o_reason[q], sizeof(int32_t) * nruns, hipMemcpyDefault, st));
    h_frob[q].resize(nruns);
    h_iters[q].resize(nruns);
    HIPCHECK(hipMemcpyAsync(h_frob[q].data(), A + o_frob[q], sizeof(float) * nruns, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(h_iters[q].data(), A + o_iters[q], sizeof(int32_t) * nruns, hipMemcpyDeviceToHost, st));
  }
  HIPCHECK(hipStreamSynchronize(st));

  ctx->obj_trace.clear();
  if (trace_stride) {
    ctx->obj_trace.resize((size_t)nunits * trace_stride);
    HIPCHECK(hipMemcpy(ctx->obj_trace.data(), A + o_trace, sizeof(double) * ctx->obj_trace.size(), hipMemcpyDeviceToHost));
    ctx->obj_trace_stride = trace_stride;
    ctx->obj_trace_nruns = nruns;
    ctx->obj_trace_unit.assign((size_t)nk * nruns, -1);
    for (int u = 0; u < nunits; ++u) ctx->obj_trace_unit[(size_t)runs[u].kidx * nruns + runs[u].ridx] = runs[u].uid;
  }

  // sse_out (Mult:125); it may be device memory: stage through a host vector.  Unweighted: normnan(X - W*H)^2.
  if (sse_out) {
    std::vector<double> all(nunits), tmp(nruns);
    if (weighted) HIPCHECK(hipMemcpy(all.data(), ctx->scratch.p + o_sse, sizeof(double) * nunits, hipMemcpyDeviceToHost));
    for (int q = 0; q < nk; ++q) {
      if (!sse_out[q]) continue;
      for (int r = 0; r < nruns; ++r) tmp[r] = (double)h_frob[q][r] * (double)h_frob[q][r];
      if (weighted)
        for (int u = 0; u < nunits; ++u)
          if (runs[u].kidx == q) tmp[runs[u].ridx] = all[u];
      HIPCHECK(hipMemcpy(sse_out[q], tmp.data(), sizeof(double) * nruns, hipMemcpyDefault));
    }
  }

  // Profile (nmfk_get_profile):
  //  "mu_loop"            wall time of the whole MU loop on the GPU (all rank groups, concurrent streams) and the
  //                        algorithmic flops of every half-step executed in it: 4*n*m*k per active unit per half-step
  //  "h_step<kp>" / "w_step<kp>"  sampled launches (every 50th iteration) of one rank group, timed on their own
  //                        stream while the other groups keep running; flops of exactly the units still active
  if (ctx->profiling) {
    float loop_ms = 0.f;
    if (hipEventElapsedTime(&loop_ms, loop_t0, loop_t1) == hipSuccess) {
      auto &E = ctx->prof["mu_loop"];
      E.ms += loop_ms;
      E.launches += 1;
      for (int q = 0; q < nk; ++q)
        for (int r = 0; r < nruns; ++r)
          E.flops += 4.0 * n * (double)m * ks[q] * (double)h_iters[q][r] * ((P.Hfixed ? 0 : 1) + (P.Wfixed ? 0 : 1));
    }
    for (const Sample &sm : prof.samples) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->events[sm.e0], ctx->events[sm.e1]) != hipSuccess) continue;
      const Group &G = groups0[sm.group];
      double active_k = 0;  // sum of the ranks of the units still iterating (positions of the work list as of the launch -> units)
      for (int u = sm.u0; u < sm.u0 + sm.cnt; ++u) {
        const NmfkRun &ru = runs0[(size_t)order_hist[(size_t)sm.epoch][(size_t)u]];
        active_k += sm.it < h_iters[ru.kidx][ru.ridx] ? ru.k : 0;
      }
      char name[64];
      if (G.kp == 0 && G.hyb)  // mixed-rank group on the split-operand MFMA kernel
        snprintf(name, sizeof(name), "%s<mfma>", sm.kind == PK_HSTEP ? "h_step" : "w_step");
      else
        snprintf(name, sizeof(name), "%s<%d>", sm.kind == PK_HSTEP ? "h_step" : "w_step", G.kp);
      auto &E = ctx->prof[name];
      E.ms += ms;
      E.launches += 1;
      E.flops += 4.0 * n * (double)m * active_k;  // W*H + the product with the ratio
    }
  }
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_mu_batch(nmfk_ctx *ctx, int k, int nruns, const float *Winit, const float *Hinit,
                              const uint64_t *seeds, const nmfk_mu_params *params, float *W_out, float *H_out,
                              float *frob_out, double *sse_out, int32_t *iters_out, int32_t *reason_out) {
  const int32_t ks[1] = {k};
  const float *wi[1] = {Winit}, *hi[1] = {Hinit};
  float *wo[1] = {W_out}, *ho[1] = {H_out}, *fo[1] = {frob_out};
  double *so[1] = {sse_out};
  int32_t *io[1] = {iters_out}, *ro[1] = {reason_out};
  return nmfk_mu_sweep(ctx, 1, ks, nruns, wi, hi, seeds, params, wo, ho, fo, so, io, ro);
}

// --------------------------------------------------------------------------------------------------------
// robustness
// --------------------------------------------------------------------------------------------------------
NMFK_EXPORT int nmfk_cluster_silhouette(nmfk_ctx *ctx, int k, int nsol, int64_t m64, const float *Hstack,
                                        int32_t *labels, float *centroids, float *point_sil, float *cluster_sil) {
  if (!ctx || !Hstack || !labels || !centroids || !point_sil || !cluster_sil)
    return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (k < 1 || nsol < 1 || m64 < 1) return fail(NMFK_ERR_BAD_ARG, "k, nsol and m must be positive");
  if (k > NMFK_MAX_K) return fail(NMFK_ERR_UNSUPPORTED, "k exceeds NMFK_MAX_K (64)");
  HIPCHECK(hipSetDevice(ctx->device));
  const int m = (int)m64;
  const size_t nT = (size_t)k * nsol;
  Bump B;
  const size_t oH = B.take(sizeof(float) * nT * m);
  const size_t oCent = B.take(sizeof(float) * (size_t)k * (m + 1));
  const size_t oLab = B.take(sizeof(int32_t) * nT);
  const size_t oCen = B.take(sizeof(float) * (size_t)k * m);
  const size_t oZ = B.take(sizeof(float) * nT * m);
  const size_t oNorm = B.take(sizeof(float) * nT);
  const size_t oD = B.take(sizeof(float) * nT * nT);
  const size_t oPs = B.take(sizeof(float) * nT);
  const size_t oCs = B.take(sizeof(float) * k);
  const size_t oFix = B.take(sizeof(int32_t));
  if (ctx->scratch.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory (cluster workspace)");
  char *S = ctx->scratch.p;
  hipStream_t st = ctx->stream;
  HIPCHECK(hipMemcpyAsync(S + oH, Hstack, sizeof(float) * nT * m, hipMemcpyDefault, st));
  nmfk_launch_cluster(k, nsol, m, (const float *)(S + oH), (float *)(S + oCent), (int32_t *)(S + oLab),
                      (float *)(S + oCen), (int32_t *)(S + oFix), st);
  nmfk_launch_silhouette(k, nsol, m, (const float *)(S + oH), (const int32_t *)(S + oLab), (float *)(S + oZ),
                         (float *)(S + oNorm), (float *)(S + oD), (float *)(S + oPs), (float *)(S + oCs), st);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(labels, S + oLab, sizeof(int32_t) * nT, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(centroids, S + oCen, sizeof(float) * (size_t)k * m, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(point_sil, S + oPs, sizeof(float) * nT, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(cluster_sil, S + oCs, sizeof(float) * k, hipMemcpyDefault, st));
  HIPCHECK(hipStreamSynchronize(st));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_silhouette(nmfk_ctx *ctx, int k, int nsol, int64_t m64, const float *stack, const int32_t *labels,
                                float *point_sil, float *cluster_sil) {
  if (!ctx || !stack || !labels || !point_sil || !cluster_sil) return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (k < 1 || nsol < 1 || m64 < 1) return fail(NMFK_ERR_BAD_ARG, "k, nsol and m must be positive");
  if (k > NMFK_MAX_K) return fail(NMFK_ERR_UNSUPPORTED, "k exceeds NMFK_MAX_K (64)");
  HIPCHECK(hipSetDevice(ctx->device));
  const int m = (int)m64;
  const size_t nT = (size_t)k * nsol;
  Bump B;
  const size_t oH = B.take(sizeof(float) * nT * m), oLab = B.take(sizeof(int32_t) * nT);
  const size_t oZ = B.take(sizeof(float) * nT * m), oNorm = B.take(sizeof(float) * nT);
  const size_t oD = B.take(sizeof(float) * nT * nT), oPs = B.take(sizeof(float) * nT), oCs = B.take(sizeof(float) * k);
  if (ctx->scratch.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory (silhouette workspace)");
  char *S = ctx->scratch.p;
  hipStream_t st = ctx->stream;
  HIPCHECK(hipMemcpyAsync(S + oH, stack, sizeof(float) * nT * m, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(S + oLab, labels, sizeof(int32_t) * nT, hipMemcpyDefault, st));
  nmfk_launch_silhouette(k, nsol, m, (const float *)(S + oH), (const int32_t *)(S + oLab), (float *)(S + oZ),
                         (float *)(S + oNorm), (float *)(S + oD), (float *)(S + oPs), (float *)(S + oCs), st);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(point_sil, S + oPs, sizeof(float) * nT, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(cluster_sil, S + oCs, sizeof(float) * k, hipMemcpyDefault, st));
  HIPCHECK(hipStreamSynchronize(st));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_cluster_stats(nmfk_ctx *ctx, int k, int nsol, int64_t n64, int64_t m64, const float *Wstack,
                                   const float *Hstack, const int32_t *labels, float *Wmean, float *Hmean,
                                   float *Wvar, float *Hvar) {
  if (!ctx || !Wstack || !Hstack || !labels || !Wmean || !Hmean || !Wvar || !Hvar)
    return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (k < 1 || nsol < 1 || n64 < 1 || m64 < 1) return fail(NMFK_ERR_BAD_ARG, "sizes must be positive");
  HIPCHECK(hipSetDevice(ctx->device));
  const int n = (int)n64, m = (int)m64;
  Bump B;
  const size_t oW = B.take(sizeof(float) * (size_t)nsol * n * k);
  const size_t oH = B.take(sizeof(float) * (size_t)nsol * m * k);
  const size_t oL = B.take(sizeof(int32_t) * (size_t)nsol * k);
  const size_t oWm = B.take(sizeof(float) * (size_t)n * k), oWv = B.take(sizeof(float) * (size_t)n * k);
  const size_t oHm = B.take(sizeof(float) * (size_t)m * k), oHv = B.take(sizeof(float) * (size_t)m * k);
  if (ctx->scratch.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory (stats workspace)");
  char *S = ctx->scratch.p;
  hipStream_t st = ctx->stream;
  HIPCHECK(hipMemcpyAsync(S + oW, Wstack, sizeof(float) * (size_t)nsol * n * k, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(S + oH, Hstack, sizeof(float) * (size_t)nsol * m * k, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(S + oL, labels, sizeof(int32_t) * (size_t)nsol * k, hipMemcpyDefault, st));
  nmfk_launch_cluster_stats(k, nsol, n, m, (const float *)(S + oW), (const float *)(S + oH), (const int32_t *)(S + oL),
                            (float *)(S + oWm), (float *)(S + oHm), (float *)(S + oWv), (float *)(S + oHv), st);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(Wmean, S + oWm, sizeof(float) * (size_t)n * k, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(Wvar, S + oWv, sizeof(float) * (size_t)n * k, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(Hmean, S + oHm, sizeof(float) * (size_t)m * k, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(Hvar, S + oHv, sizeof(float) * (size_t)m * k, hipMemcpyDefault, st));
  HIPCHECK(hipStreamSynchronize(st));
  return NMFK_OK;
}

NMFK_EXPORT int nmfk_frobenius(nmfk_ctx *ctx, int k, const float *W, const float *H, double *out) {
  if (!ctx || !W || !H || !out) return fail(NMFK_ERR_BAD_ARG, "null argument");
  if (!ctx->Xc && !ctx->sparse) return fail(NMFK_ERR_NO_X, "nmfk_set_X has not been called");
  if (k < 1) return fail(NMFK_ERR_BAD_ARG, "k must be >= 1");
  HIPCHECK(hipSetDevice(ctx->device));
  const int n = (int)ctx->n, m = (int)ctx->m;
  const int tiles = (n + 255) / 256;
  hipStream_t st = ctx->stream;
  if (ctx->sparse) {
    // one-unit arena: the init kernel turns (W, H) into the signal-major layout, then the sparse objective
    if (k > NMFK_MAX_K) return fail(NMFK_ERR_UNSUPPORTED, "k exceeds NMFK_MAX_K (64)");
    const int kp = nmfk_padded_k(k);
    Bump B;
    const size_t oRun = B.take(sizeof(NmfkRun)), oSt = B.take(sizeof(NmfkState)), oPtr = B.take(2 * sizeof(void *));
    const size_t oFlag = B.take(64), oOut = B.take(sizeof(double));
    const size_t oWi = B.take(sizeof(float) * (size_t)n * k), oHi = B.take(sizeof(float) * (size_t)m * k);
    NmfkRun rd;
    memset(&rd, 0, sizeof(rd));
    rd.k = k;
    rd.kp = kp;
    rd.nsW = rd.nsH = 1;
    rd.oWt = (int64_t)B.take(sizeof(float) * (size_t)n * kp);
    rd.oH0 = rd.oH1 = (int64_t)B.take(sizeof(float) * (size_t)m * kp);
    rd.osumW = (int64_t)B.take(sizeof(double) * kp);
    rd.osumH = (int64_t)B.take(sizeof(double) * kp);
    rd.ossepart = (int64_t)B.take(sizeof(double) * (tiles + 1));
    rd.ogram = (int64_t)B.take(sizeof(double) * nmfk_gram_doubles(n, m, kp));
    if (ctx->scratch.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory");
    char *S = ctx->scratch.p;
    void *ptrs[2] = {S + oWi, S + oHi};
    HIPCHECK(hipMemcpyAsync(S + oRun, &rd, sizeof(rd), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S + oPtr, ptrs, sizeof(ptrs), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S + oWi, W, sizeof(float) * (size_t)n * k, hipMemcpyDefault, st));
    HIPCHECK(hipMemcpyAsync(S + oHi, H, sizeof(float) * (size_t)m * k, hipMemcpyDefault, st));
    HIPCHECK(hipMemsetAsync(S + oFlag, 0, 64, st));
    HIPCHECK(hipStreamSynchronize(st));  // rd / ptrs are stack objects
    NmfkInitArgs ia;
    ia.arena = S;
    ia.n = n;
    ia.m = m;
    ia.runs = (const NmfkRun *)(S + oRun);
    ia.state = (NmfkState *)(S + oSt);
    ia.nunits = 1;
    ia.Winit = (const float *const *)(S + oPtr);
    ia.Hinit = (const float *const *)(S + oPtr) + 1;
    ia.PW = 1;
    ia.PH = 1;
    ia.nan_flag = (int32_t *)(S + oFlag);
    nmfk_launch_init_f32(ia, st);
    NmfkSparseArgs sp;
    sp.objw = 0.0;
    sp.ntile_obj = 0;
    sp.clampw = 0;
    sp.arena = S;
    sp.ptr = ctx->rowptr;
    sp.rec = ctx->rec_csr;
    sp.nrec = (int32_t)std::max<int64_t>(ctx->nnz, 1);
    sp.runs = ia.runs;
    sp.state = ia.state;
    sp.L = n;
    sp.which = 1;
    sp.it = 0;
    sp.PW = sp.PH = 1;
    sp.force = 1;
    sp.split = 0;
    sp.ell = nullptr;
    sp.ellptr = nullptr;
    sp.ngb = 0;
    sp.D = m;
    nmfk_launch_sp_obj_f32(&sp, n, m, 0, 0, 1.0, 0, 1, st);
    nmfk_launch_sum_parts_f32(S, ia.runs, 1, tiles + 1, (double *)(S + oOut), st);
    HIPCHECK(hipGetLastError());
    double v = 0;
    HIPCHECK(hipMemcpyAsync(&v, S + oOut, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    *out = sqrt(v > 0 ? v : 0.0);
    return NMFK_OK;
  }
  Bump B;
  const size_t oW = B.take(sizeof(float) * (size_t)n * k), oH = B.take(sizeof(float) * (size_t)m * k);
  const size_t oP = B.take(sizeof(double) * tiles);
  if (ctx->scratch.ensure(B.off)) return fail(NMFK_ERR_HIP, "out of device memory");
  char *S = ctx->scratch.p;
  HIPCHECK(hipMemcpyAsync(S + oW, W, sizeof(float) * (size_t)n * k, hipMemcpyDefault, st));
  HIPCHECK(hipMemcpyAsync(S + oH, H, sizeof(float) * (size_t)m * k, hipMemcpyDefault, st));
  nmfk_launch_frob(ctx->Xc, n, m, k, (const float *)(S + oW), (const float *)(S + oH), (double *)(S + oP), st);
  HIPCHECK(hipGetLastError());
  std::vector<double> part(tiles);
  HIPCHECK(hipMemcpyAsync(part.data(), S + oP, sizeof(double) * tiles, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipStreamSynchronize(st));
  double s = 0;
  for (int t = 0; t < tiles; ++t) s += part[t];
  *out = sqrt(s);
  return NMFK_OK;
}
