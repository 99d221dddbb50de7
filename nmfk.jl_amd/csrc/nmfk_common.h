// Shared host/device declarations of libnmfk_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

// One unit of the flat work list = one restart of one rank k: "a factorization".
// Internal factor layout (both factors "signal-major", so the two half-steps are the same kernel):
//   Wt : kp x n   element (a, i) at Wt[a + i*kp]      (W transposed: row i of W contiguous)
//   H  : kp x m   element (a, j) at H [a + j*kp]      (double-buffered: H0, H1)
// kp = k for k <= 16, else k padded up to the next instantiated width (padding rows are zero).
struct NmfkRun {
  int32_t k;         // true rank
  int32_t kp;        // padded rank = leading dimension of Wt / H
  int32_t kidx;      // index of the rank in the caller's ks[]
  int32_t ridx;      // restart index within the rank
  // byte offsets into the sweep arena (kernels form `arena + offset`, so the compiler knows the pointers are
  // global memory and can use scalar loads for wave-uniform reads)
  int64_t oWt;       // T[kp*n]
  int64_t oH0, oH1;  // T[kp*m] x2, double-buffered by iteration parity (equal when Hfixed)
  int64_t opart;     // T[max(Sh*kp*m, Sw*kp*n)] partial numerators of the current half-step
  int64_t osumW;     // double[PW][kp] colsum(W) as PW partial vectors (denominator of the H half-step, Mult:67)
  int64_t osumH;     // double[PH][kp] rowsum(H) as PH partial vectors (denominator of the W half-step, Mult:70)
  int64_t ossepart;  // double[ntile_n] per-workgroup partial objective
  int64_t ocanon;    // int32[2 m]: canonical co-clustering partition of the previous check (Mult:101-116), then check_b's index scratch
  uint64_t seed;
  int32_t nsW, nsH;  // slots of the sum tables this unit's kernels write (<= PW, PH); the others stay zero
  // split-operand MFMA half-step (nmfk_step_hyb.hip): hyb = split width KS (8 or 16; 0 = unit does not use it)
  int32_t hyb;
  int32_t uid;       // position of the unit in the work list the sweep started with (the retire-aware schedule reorders the list)
  int64_t ogram;     // sparse X: double[nmfk_gram_doubles()] partial Gram matrices of the factors (objective)
  int64_t osnapW;    // float[16]: colsum(W) as the H half-step read it, kept for the W half-step that sums the H partials itself
                     // (NmfkStepArgs::fuse_red; its own workgroups overwrite osumW while others of the unit still start)
};

// sparse objective: <W'W, HH'> from partial Gram matrices over chunks of NMFK_GRAM_ROWS factor rows; a chunk's rows
// are split once more (rp parts) when the matrix is a single 16 x 16 block, so that the four waves all have work
#define NMFK_GRAM_ROWS 4096
static inline int nmfk_gram_nb(int kp) { return (kp + 15) / 16; }
static inline int nmfk_gram_rp(int kp) { return nmfk_gram_nb(kp) * nmfk_gram_nb(kp) >= 4 ? 1 : 4; }
static inline size_t nmfk_gram_doubles(int64_t n, int64_t m, int kp) {
  const size_t chunks = (size_t)((n + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS + (m + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS);
  return chunks * nmfk_gram_rp(kp) * nmfk_gram_nb(kp) * nmfk_gram_nb(kp) * 256;
}

// Stop-rule state machine of NMFmultiplicative (Mult:57-63), one per unit, device resident.
struct NmfkState {
  double best;       // objvalue_best
  double last_obj;   // last monitored objective
  int32_t iters;     // iterations executed (valid once inactive)
  int32_t baditers;
  int32_t reattempts;
  int32_t inc;
  int32_t have_old;  // consold is not the initial falses(m,m)
  int32_t active;    // still inside the while loop
  int32_t reason;    // NMFK_STOP_*
  int32_t lowflag;   // 1: the clamp of the next check block (Mult:99-100) has to look at the unit's factors.  Set by init; units whose
                     // half-step kernels watch the values they write in a check iteration (NmfkCheckArgs::track_low) have it cleared
                     // by check_b and set again by a fused finish that writes a value below eps()
};

// Arguments of one half-step over all units.  lane dimension L (contiguous in the X copy used), loop
// dimension D:  H half-step: L = m, D = n, X copy = row-major (Xr);  W half-step: L = n, D = m, X copy =
// column-major (Xc).
struct NmfkStepArgs {
  char *arena;
  const float *X;   // element (l, d) at X[l + d*ld]
  const float *Xalt;  // the other copy: element (l, d) at Xalt[d + l*D] (loop dimension contiguous; MFMA variant)
  const float *Xtile; // Xalt in 16 x 16 blocks in MFMA lane order (nmfk_step_hyb.hip), or null
  int64_t ld;
  int32_t L, D;
  int32_t S;        // grid-level splits of the loop dimension (S > 1 => fused = 0, reduce kernel finishes)
  int32_t wsplit;   // 1: each wave owns 64*LB lane elements; 4 / 8: that many waves share them and split the loop range
  int32_t fused;    // the step kernel finishes the update itself
  int32_t PW, PH;   // slots of the sum tables of W and H
  int32_t dchunk;   // loop extent per split
  int32_t which;    // 0 = H half-step, 1 = W half-step
  int32_t it;       // 0-based iteration index: reads H(it&1), writes H((it+1)&1)
  int32_t has_nan;  // X holds NaN (missing) entries: EM imputation semantics of Mult:72
  float lambda;
  const NmfkRun *runs;
  NmfkState *state;
  int32_t nunits;
  int32_t force;    // ignore the active flags
  int32_t res_wgs;  // > 0: the split-operand MFMA units run the RESIDENT form of this half-step (nmfk_step_hyb.hip) with
                    // this many workgroups per unit (= sum-table slots they write); 0: the streaming form
  int32_t lag;      // split-operand streaming form: which half-step instantiation a launch takes -- < 0: the lagged one where a wave walks 32 chunks
                    // or more (hyb_step_body, LAG), 0: never, > 0: always (NMFK_HYB_LAG; the two give the same bits)
  int32_t fuse_red; // W half-step, resident form: > 0 = the H half-step of this iteration left S = fuse_red partial numerators per unit
                    // (its loop range was split over workgroups) and NO reduce_kernel has run: every workgroup of this launch sums
                    // them while it stages H -- H_new = H .* sum(partials) ./ colsum(W), Mult:67, reduce_kernel's arithmetic --,
                    // workgroup 0 of a unit also writes H_new and rowsum(H) (round 5: the reduce launch was 15 % of a 60-unit share).
                    // The H half-step carries the same value: its workgroup 0 copies colsum(W) to NmfkRun::osnapW
  int32_t bnum;     // wide ranks (wide2_step_kernel): the numerators run on the bf16 matrix pipe from exact three-term splits of the ratios
                    // (round 6; NMFK_WIDE_BN): 1 (default) = at 48 and 64 signals, 2 = at 32 too, 0 = never (fp32 matrix pipe, rounds 3-5)
  int32_t clampw;   // > 0: the W half-step's fused finishes of a check iteration write max(W, eps()) themselves (Mult:100; the deferred
                    // check of nmfk_mu_sweep: nothing reads W between the half-step and the clamp, the pass then only walks H).
                    // The value is the sweep's maxiter: the check AT maxiter is not deferred (no half-step follows), its objective
                    // is taken BEFORE the clamp as in the reference (Mult:74, 99-100) -- iterations it + 1 < clampw only.
};

struct NmfkSseArgs {
  char *arena;
  const float *Xc;  // column-major copy, element (i, j) at Xc[i + j*n]
  const float *Xr;  // row-major copy, element (i, j) at Xr[j + i*m] (the MFMA objective kernel of ranks > 16)
  const float *Wgt; // optional n x m weight array (column-major) of the monitored objective (Mult:74), or null
  int32_t n, m;
  int32_t hsel;     // which H buffer parity to read; -1 => per-unit final buffer (state.iters&1)
  double weight;
  const NmfkRun *runs;
  NmfkState *state;
  int32_t nunits;
  int32_t force;        // finish pass: every unit, NaN residuals skipped (normnan)
  int32_t total_iters;  // iterations the host loop executed (selects the final H buffer of still-active units)
};

// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device: the one-process multi-GPU form
// (nmfk_multi_*) launches the same kernel on several devices from several threads.  `done`: one bit per device.
static inline void nmfk_allow_dynamic_lds(const void *kernel, std::atomic<uint64_t> &done, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}

// half-step / objective on sparse X: CSC view for the H half-step (L = m), CSR view otherwise (L = n)
struct NmfkSparseArgs {
  char *arena;
  const int32_t *ptr;   // CSC colptr (H half-step) or CSR rowptr (W half-step), length L + 1
  const int2 *rec;      // the non-zeros as records (.x = row / column index, .y = the fp32 value's bits), >= 1 entry
  int32_t nrec;         // number of records allocated (>= 1): loads past a lane element's range are clamped to it
  const NmfkRun *runs;
  const NmfkState *state;
  int32_t L, which, it, PW, PH, force;
  int32_t split;  // half-step: 1 = a workgroup walks ONE pass of its tile (NMFK_TILE / LPR lane elements) and owns a
                  // sum-table slot of that size -- LPR x the workgroups (the H half-step has few lane elements)
  // blocked form (round 3, sp_blk_kernel): the non-zeros as sliced ELL -- slices of 64 lane elements (one per lane of a
  // wave) x granules of NMFK_SPB_ROWS rows of the gathered factor; the run of (slice s, granule b) starts at slot row
  // ellptr[s * ngb + b] and is ellptr[.. + 1] - ellptr[..] slot rows long (the longest lane element of the slice in that
  // granule); slot row r holds 64 records ell[64 r + lane], .x = -1 where a lane element has fewer.  Null = the gather form.
  const int2 *ell;
  const int32_t *ellptr;
  int32_t ngb;
  int32_t D;      // the loop dimension (rows of the gathered factor)
  // deferred check (nmfk_mu_sweep; blocked form): objw > 0: the H half-step also leaves the non-zero terms of the objective of the
  // factors it reads, scaled by objw^2, in ossepart[1 + tile] (ntile_obj entries behind slot 0 in all: the rest zeroed);
  // clampw (> 0: the sweep's maxiter, see NmfkStepArgs::clampw): the W half-step's finish of a check iteration before the last writes max(W, eps())
  double objw;
  int32_t ntile_obj, clampw;
};
#define NMFK_SPB_ROWS 1024  // blocked form: lane elements per workgroup (one per thread; = its sum-table slot) and rows of the
                            // gathered factor per granule of the sliced ELL
#define NMFK_SPB_LDS (160 * 1024)
// ranks the blocked form serves: up to 32 signals (a lane holds its lane element's row and numerators in registers, and
// 1024 rows of the gathered factor fit in LDS).  NMFK_SPB_MINK (a build-time knob of the A/B runs in profiles/r03) keeps
// the ranks up to it in the gather form.
#ifndef NMFK_SPB_MINK
#define NMFK_SPB_MINK 0
#endif
__host__ __device__ static inline int nmfk_sp_blk_rank(int kp) { return kp > NMFK_SPB_MINK && kp <= 32; }
// words between the staged rows of 4 * nc signals.  A ds_read_b128 is served 16 lanes at a time, each lane on a window of four banks: window =
// (row * stride / 4 + chunk) mod 16.  Round 6: the stride / 4 is ODD for every nc (nc + 1 for even nc, nc + 2 for odd), so that the rows map onto all 16
// windows; rounds 3-5 used nc + 1 throughout -- at 12 signals (stride 16 words = 64 B) the rows of a read fell on FOUR windows (a 4-way conflict on every
// gather of the ranks 9..12), at 4 and 20 signals on eight.  (nc = 1 keeps 8 words: 12 would cost it two of its five granules per stage.)
static inline int nmfk_spb_stride(int nc) { return nc == 1 ? 8 : (nc & 1) ? 4 * nc + 8 : 4 * nc + 4; }
// granules of the gathered factor staged at a time
static inline int nmfk_spb_gps(int nc) { return NMFK_SPB_LDS / (nmfk_spb_stride(nc) * 4) / NMFK_SPB_ROWS; }
// lanes per lane element of the sparse kernels (four signals each) and lane elements per sum-table slot
static inline int nmfk_sp_lpr(int kp) { return kp <= 4 ? 1 : kp <= 8 ? 2 : kp <= 16 ? 4 : kp <= 32 ? 8 : 16; }
static inline int nmfk_sp_slot(int kp, int split) { return split ? 256 / nmfk_sp_lpr(kp) : 256; }

struct NmfkCheckArgs {
  char *arena;
  int32_t n, m;
  int32_t it;       // index of the iteration just completed (0-based); (it+1) % 10 == 0
  int32_t ntile_n;  // number of ssepart entries per unit
  int32_t PW, PH;
  double tol, tolOF;
  int64_t maxiter;
  int32_t maxbaditers, maxreattempts, stopconv;
  const NmfkRun *runs;
  NmfkState *state;
  int32_t nunits;
  double *trace;         // optional (nmfk_set_objective_trace): monitored objective of unit u at check c -> trace[u * trace_stride + c]
  int32_t trace_stride;
  int32_t track_low;     // the units' half-step kernels maintain NmfkState::lowflag: the clamp pass skips units whose flag is clear
  int32_t w_clamped;     // W was clamped by the half-step that wrote it (NmfkStepArgs::clampw): the clamp pass walks H only
};

struct NmfkFinishArgs {
  char *arena;
  int32_t n, m;
  int32_t ntile_n;
  int32_t total_iters;  // iterations the host loop executed
  int32_t normalize;
  const NmfkRun *runs;
  NmfkState *state;
  int32_t nunits;
  float *const *Wout;   // per kidx: nruns stacked n x k (device staging)
  float *const *Hout;   // per kidx: nruns stacked k x m
  float *const *frob;   // per kidx: nruns
  int32_t *const *iters;
  int32_t *const *reason;
};

struct NmfkInitArgs {
  char *arena;
  int32_t n, m;
  const NmfkRun *runs;
  NmfkState *state;
  int32_t nunits;
  const float *const *Winit;  // per kidx (device staging) or null entries
  const float *const *Hinit;
  int32_t PW, PH;
  int32_t *nan_flag;          // set to 1 when an initial value is NaN (Mult:42-44,52-54)
};

static inline int nmfk_padded_k(int k) {
  if (k <= 16) return k;
  static const int w[] = {20, 24, 28, 32, 40, 48, 56, 64};
  for (int i = 0; i < 8; i++)
    if (k <= w[i]) return w[i];
  return -1;
}

// switch over the instantiated factor widths
#define NMFK_DISPATCH_KP(kp, CALL)                                                                    \
  switch (kp) {                                                                                       \
    case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;   case 4: CALL(4); break;   \
    case 5: CALL(5); break;   case 6: CALL(6); break;   case 7: CALL(7); break;   case 8: CALL(8); break;   \
    case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break; \
    case 20: CALL(20); break; case 24: CALL(24); break; case 28: CALL(28); break; case 32: CALL(32); break; \
    case 40: CALL(40); break; case 48: CALL(48); break; case 56: CALL(56); break; case 64: CALL(64); break; \
    default: break;                                                                                   \
  }

// offset of the H buffer of parity `par`
#define NMFK_HOFF(rd, par) (((par)&1) ? (rd).oH1 : (rd).oH0)

#ifndef NMFK_LB
#define NMFK_LB 2  // lane elements per thread for k <= 16
#endif
#ifndef NMFK_LB4_MAXK
#define NMFK_LB4_MAXK 0  // ranks up to this use 4 lane elements per thread
#endif
#define NMFK_LB_OF(KP) ((KP) <= NMFK_LB4_MAXK ? 4 : ((KP) <= 16 ? NMFK_LB : 1))
#ifndef NMFK_MERGE_MAX_RUNS
#define NMFK_MERGE_MAX_RUNS 4  // restarts per rank at or below which the ranks <= 16 share launches (see nmfk_mu_sweep)
#endif
#ifndef NMFK_MERGE_GROUPS
#define NMFK_MERGE_GROUPS 2    // number of mixed-rank launch groups then
#endif
#ifndef NMFK_XCD_REMAP
#define NMFK_XCD_REMAP 1     // wide-rank MFMA kernel: restarts of one lane tile on one XCD (L2 sharing of X at large sizes)
#endif
#ifndef NMFK_UNIT_FAST
#define NMFK_UNIT_FAST 0     // half-step grids: 1 = unit is the fast dimension (see the launchers: measured worse for L2)
#endif
#ifndef NMFK_EPACK
#define NMFK_EPACK 1         // packed VALU lanes = the two lane elements of a thread (1) or adjacent signals (0)
#endif
#ifndef NMFK_XBUF
#define NMFK_XBUF 1          // X entries through buffer loads (scalar address arithmetic); 0 = global loads
#endif
#ifndef NMFK_WITH_MERGED_F32
#define NMFK_WITH_MERGED_F32 1  // 0: build libnmfk_hip.so without the fp32 instantiation of the mixed-rank packed-VALU kernel.
                                // (It is where the gfx950 packed-fp32 hazard of DESIGN.md was first seen; with the
                                // broadcast-first operand rule of nmfk_step_impl.h its code no longer contains the unsafe
                                // instruction form -- scripts/isa_lint_pk_opsel.py, tests/test_isa_lint.py.)
#endif
#ifndef NMFK_MULTI_MAXK
#define NMFK_MULTI_MAXK 16    // widest rank served by the mixed-rank kernel (8, 12, 14 or 16); wider ranks keep their own launches
#endif
#ifndef NMFK_MULTI_MINWAVES
#define NMFK_MULTI_MINWAVES 3 // waves per SIMD requested for the mixed-rank kernel (2: 179 VGPRs, 12 % slower; 4: spills, 65 % slower)
#endif
#ifndef NMFK_MULTI_LB
#define NMFK_MULTI_LB 2      // lane elements per thread of the mixed-rank kernel (1 measured 50 % slower: the wave-uniform rows are shared)
#endif
#ifndef NMFK_WIDE_NT
#define NMFK_WIDE_NT 2   // 16-wide lane tiles per wave of the all-MFMA kernel for k > 16
#endif
#ifndef NMFK_FASTDIV
#define NMFK_FASTDIV 2  // 2: v_rcp_f32 (1 ulp) * x; 1: + one Newton step; 0: IEEE divide.  fp32 ratio X/(W*H) by v_rcp_f32 + one Newton step instead of the IEEE divide sequence
#endif
#define NMFK_TILE 256  // threads per workgroup = lane-tile width of the half-step kernels

// launchers (nmfk_step_f32.hip / nmfk_step_f64.hip)
// Units are sorted by k descending; a "group" is the contiguous range of units of one rank.  All per-group
// launches take (u0, cnt) = that range and the group's stream.
#define NMFK_DECLARE_LAUNCHERS(SUF)                                                                               \
  void nmfk_launch_init_##SUF(const NmfkInitArgs &a, hipStream_t s);                                              \
  void nmfk_launch_step_##SUF(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt,          \
                              hipStream_t s);                                                                     \
  void nmfk_launch_step_multi_##SUF(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int u0, int cnt, hipStream_t s); \
  void nmfk_launch_reduce_##SUF(const NmfkStepArgs &a, int u0, int cnt, hipStream_t s);                           \
  void nmfk_launch_sse_##SUF(const NmfkSseArgs &a, int u0, int cnt, hipStream_t s);                               \
  void nmfk_launch_check_##SUF(const NmfkCheckArgs &a, int u0, int cnt, hipStream_t s, int parts = 7);               \
  void nmfk_launch_sum_parts_##SUF(char *arena, const NmfkRun *runs, int nunits, int ntile, double *out,           \
                                   hipStream_t s);                                                                \
  void nmfk_launch_sp_step_##SUF(const void *sparse_args, int kp, int u0, int cnt, hipStream_t s);                \
  void nmfk_launch_sp_obj_##SUF(const void *sparse_args, int n, int m, int hsel, int total_iters, double weight,   \
                                int u0, int cnt, hipStream_t s, int parts = 3);                                   \
  void nmfk_launch_finish_##SUF(const NmfkFinishArgs &a, hipStream_t s);
NMFK_DECLARE_LAUNCHERS(f32)
void nmfk_launch_kmeans(const float *X, int d, int n, int k, int repeats, int maxiter, double tol, uint64_t seed,
                        int32_t *assign, float *costs, float *work, float *centers, int32_t *counts, double *total,
                        int32_t *iters, int32_t *conv, hipStream_t s);
void nmfk_launch_point_silhouettes(const float *X, int d, int n, const int32_t *assign, const int32_t *cnt, int k, float *Z,
                                   float *sil, hipStream_t s);
void nmfk_launch_step_mfma_wide_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt,
                                    hipStream_t s);
int nmfk_mfma_wide_lane_tile(int wsplit);
int nmfk_hyb_lane_tile(int wsplit);
int nmfk_wide2_ok(int kp);  // wide rank width served by the split-operand form of the all-MFMA half-step (wide2_step_kernel)
int nmfk_wide2_lane_tile();
void nmfk_launch_step_wide2_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt, hipStream_t s, double objw = 0.0);
void nmfk_launch_wide2_sse(const NmfkStepArgs &w, const NmfkStepArgs *dw, double weight, int hsel, int kp, int u0, int cnt, hipStream_t s);
int nmfk_hyb_resident_waves();  // waves per workgroup of the resident form
size_t nmfk_hyb_resident_lds(int variant, int D);  // LDS bytes of the resident form for a loop dimension D, 0 = not applicable
int nmfk_hyb_variant(int k);  // kernel variant of rank k on the split-operand MFMA half-step: 4 / 8 / 12 / 16
#if defined(__HIPCC__)
// sum over the P slots of a sum table of signal `idx` (entry pp * stride + idx), in slot order, NB loads in flight: one load at a
// time is a chain of P memory round trips in front of every fused finish (wide2_step_kernel's note has the measurement).
template <int NB>
__device__ __forceinline__ double nmfk_slot_sum(const double *__restrict__ tab, int stride, int P, int idx) {
  double sd = 0;
  for (int p0 = 0; p0 < P; p0 += NB) {
    double sv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) sv[j] = p0 + j < P ? tab[(p0 + j) * stride + idx] : 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (p0 + j < P) sd += sv[j];
  }
  return sd;
}
#endif

void nmfk_launch_step_hyb_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int ks, int u0, int cnt, hipStream_t s, double objw = 0.0);
int nmfk_hyb_step_parts(const NmfkStepArgs &a);
void nmfk_launch_hyb_sse(const NmfkStepArgs &w, const NmfkStepArgs *dw, double weight, int hsel, int ks, int u0, int cnt,
                         hipStream_t s);
void nmfk_launch_hyb_tile(const float *src, int L, int D, float *out, hipStream_t s);
void nmfk_launch_sse_mfma_wide_f32(const NmfkSseArgs &a, int kp, int u0, int cnt, hipStream_t s);
NMFK_DECLARE_LAUNCHERS(f64)

// nmfk_cluster.hip
void nmfk_launch_preprocess(const float *Xin, int64_t ldx, int64_t n, int64_t m, float lambda, float *Xc, float *Xr,
                            unsigned long long *counts /* [0]=neg [1]=nan [2]=zero */, hipStream_t s);
void nmfk_launch_fill_uniform(uint64_t seed, uint64_t offset, int64_t count, float *out, hipStream_t s);
void nmfk_launch_cluster(int k, int nsol, int m, const float *Hstack, float *work, int32_t *labels, float *centroids,
                         int32_t *needfix, hipStream_t s);
void nmfk_launch_silhouette(int k, int nsol, int m, const float *Hstack, const int32_t *labels, float *Z, float *norms,
                            float *D, float *psil, float *csil, hipStream_t s);
void nmfk_launch_cluster_stats(int k, int nsol, int n, int m, const float *Wstack, const float *Hstack,
                               const int32_t *labels, float *Wmean, float *Hmean, float *Wvar, float *Hvar,
                               hipStream_t s);
void nmfk_launch_frob(const float *Xc, int n, int m, int k, const float *W, const float *H, double *partial,
                      hipStream_t s);
