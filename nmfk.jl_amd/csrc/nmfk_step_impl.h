// Multiplicative-update kernels of libnmfk_hip, written for gfx950 (wave64, 256 CUs).
// Included by nmfk_step_f32.hip / nmfk_step_f64.hip with
//   NMFK_T   = float | double   arithmetic type of W, H and W*H
//   NMFK_SUF = f32 | f64
//
// The reference's iteration (src/NMFkMultiplicative.jl:64-72)
//     H = H .* (W' * (X ./ (W*H))) ./ sum(W;dims=1)'          (67)
//     W = W .* ((X ./ (W*H)) * H') ./ sum(H;dims=2)'          (70)
// is two passes over X that never materialise W*H or X./(W*H).  Because both factors are stored
// signal-major (Wt: k x n, H: k x m) the two half-steps are ONE kernel applied to X' and X:
//     lane factor A (k values per lane, in VGPRs), loop factor B (one k-vector per loop step, wave-uniform,
//     fetched with scalar loads into SGPRs), numerator acc[k] in VGPRs:
//         p = <a, b_d>;  q = x[l,d] / p;  acc += q * b_d
// X is read with unit stride across the wave in both half-steps (row-major copy for H, column-major for W).
#include "nmfk_common.h"
#include "../../include/nmfk_hip.h"
#include "nmfk_rng.h"
#include <algorithm>
#include <type_traits>

#define NMFK_CAT2(a, b) a##b
#define NMFK_CAT(a, b) NMFK_CAT2(a, b)
#define NMFK_NAME(base) NMFK_CAT(NMFK_CAT(base, _), NMFK_SUF)

namespace {

typedef NMFK_T T;

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
#define NMFK_PTR(TYPE, g, off) ((TYPE *)((g).arena + (off)))

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 256 threads of a workgroup, fixed order => bitwise reproducible.  Result valid in thread 0
// (and broadcast through sh[4]).  sh: >= 5 doubles.
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return sh[4];
}

// sum table of a factor: P slots of kp doubles; consumers add the slots in order.  The half-step kernels write one
// slot per lane tile; the grid-parallel helper kernels (init, clamp, reduce) split the lane range into P equal
// chunks, workgroup b owning chunk b and slot b.  Any partition works as long as every slot is written or zero:
// a unit uses the first NmfkRun::nsW / nsH slots (what its fused half-step kernel writes), the rest stay zero.
__device__ __forceinline__ void slot_range(int len, int P, int b, int &l0, int &l1) {
  const int ch = (len + P - 1) / P;
  l0 = min(len, b * ch);
  l1 = min(len, l0 + ch);
}

__device__ __forceinline__ void zero_slot(double *slot, int kp) {
  if ((int)threadIdx.x < kp) slot[threadIdx.x] = 0.0;
}

// out[c] = sum_{l in [l0,l1)} F[c + l*kp]  (c < k; zero for the padding signals), fixed order.  Thread (c, r) adds
// the elements l = l0 + r, l0 + r + rows, ... so that consecutive threads read consecutive addresses.
// sh: NMFK_TILE doubles.  out may be LDS or global memory.
__device__ __forceinline__ void range_signal_sums(const T *F, int kp, int k, int l0, int l1, double *out, double *sh) {
  const int rows = NMFK_TILE / kp;
  const int tid = threadIdx.x, c = tid % kp, r = tid / kp;
  double s = 0;
  if (r < rows)
    for (int l = l0 + r; l < l1; l += rows) s += (double)F[c + (int64_t)l * kp];
  __syncthreads();
  if (r < rows) sh[tid] = s;
  __syncthreads();
  if (tid < kp) {
    double t = 0;
    for (int q = 0; q < rows; ++q) t += sh[q * kp + tid];
    out[tid] = tid < k ? t : 0.0;
  }
  __syncthreads();
}

// the same with the clamp of Mult:99-100 in the one pass: F[c + l*kp] = max(F, eps) for the signals c < k (a NaN stays), then the sums of
// the clamped values in the order range_signal_sums adds them (the same bits as clamping first and summing afterwards)
__device__ __forceinline__ void range_clamp_sums(T *F, int kp, int k, int l0, int l1, T eps, double *out, double *sh) {
  const int rows = NMFK_TILE / kp;
  const int tid = threadIdx.x, c = tid % kp, r = tid / kp;
  double s = 0;
  if (r < rows)
    for (int l = l0 + r; l < l1; l += rows) {
      T v = F[c + (int64_t)l * kp];
      if (c < k && v < eps) {
        v = eps;
        F[c + (int64_t)l * kp] = v;
      }
      s += (double)v;
    }
  __syncthreads();
  if (r < rows) sh[tid] = s;
  __syncthreads();
  if (tid < kp) {
    double t = 0;
    for (int q = 0; q < rows; ++q) t += sh[q * kp + tid];
    out[tid] = tid < k ? t : 0.0;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------
// init: W = rand(n,k) then H = rand(k,m) (Mult:38,48) or the caller's Winit/Hinit (Mult:40-41,50-51);
// state of Mult:57-63; colsum(W), rowsum(H).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void init_kernel(NmfkInitArgs g) {  // grid (max(PW, PH), units)
  __shared__ double sh[NMFK_TILE];
  const int u = blockIdx.y, b = blockIdx.x;
  const NmfkRun rd = g.runs[u];
  const int k = rd.k, kp = rd.kp, n = g.n, m = g.m;
  T *Wt = NMFK_PTR(T, g, rd.oWt);
  T *H = NMFK_PTR(T, g, rd.oH0);
  const float *Wi = g.Winit ? g.Winit[rd.kidx] : nullptr;
  const float *Hi = g.Hinit ? g.Hinit[rd.kidx] : nullptr;
  const uint64_t key = nmfk_splitmix64(rd.seed);
  int bad = 0;
  if (b >= rd.nsW && b < g.PW) zero_slot(NMFK_PTR(double, g, rd.osumW) + (int64_t)b * kp, kp);
  if (b >= rd.nsH && b < g.PH) zero_slot(NMFK_PTR(double, g, rd.osumH) + (int64_t)b * kp, kp);
  if (b < rd.nsW) {
    int l0, l1;
    slot_range(n, rd.nsW, b, l0, l1);
    for (int64_t e = (int64_t)l0 * kp + threadIdx.x; e < (int64_t)l1 * kp; e += NMFK_TILE) {
      const int c = (int)(e % kp);
      const int64_t i = e / kp;
      float v = 0.f;
      if (c < k) v = Wi ? Wi[(int64_t)rd.ridx * n * k + i + (int64_t)c * n] : nmfk_uniform_keyed(key, (uint64_t)(i + (int64_t)c * n));
      bad |= (v != v);
      Wt[e] = (T)v;
    }
    __syncthreads();
    range_signal_sums(Wt, kp, k, l0, l1, NMFK_PTR(double, g, rd.osumW) + (int64_t)b * kp, sh);
  }
  if (b < rd.nsH) {
    int l0, l1;
    slot_range(m, rd.nsH, b, l0, l1);
    for (int64_t e = (int64_t)l0 * kp + threadIdx.x; e < (int64_t)l1 * kp; e += NMFK_TILE) {
      const int c = (int)(e % kp);
      const int64_t j = e / kp;
      float v = 0.f;
      if (c < k)
        v = Hi ? Hi[(int64_t)rd.ridx * m * k + c + j * k] : nmfk_uniform_keyed(key, (uint64_t)((int64_t)n * k + c + j * k));
      bad |= (v != v);
      H[e] = (T)v;
    }
    __syncthreads();
    range_signal_sums(H, kp, k, l0, l1, NMFK_PTR(double, g, rd.osumH) + (int64_t)b * kp, sh);
  }
  if (bad) atomicOr(g.nan_flag, 1);
  if (b == 0 && threadIdx.x == 0) {
    NmfkState s;
    s.best = __builtin_inf();
    s.last_obj = __builtin_nan("");
    s.iters = 0;
    s.baditers = 0;
    s.reattempts = 0;
    s.inc = 0;
    s.have_old = 0;
    s.active = 1;
    s.reason = 0;
    s.lowflag = 1;
    g.state[u] = s;
  }
}

// ------------------------------------------------------------------------------------------------------
// half-step (the hot kernel)
//
// Thread layout: a workgroup owns LPW*LB consecutive lane elements (columns of X for the H half-step, rows
// for the W half-step); LPW = 256 lane slots (wsplit = 1: every wave owns its own 64*LB elements and walks
// the whole loop range) or 64 (wsplit = 4: the four waves share 64*LB elements and each walks a quarter of
// the loop range; their numerators are summed through LDS in a fixed order).  LB elements per thread share
// every wave-uniform loop-factor row (SGPRs), so scalar traffic and loop overhead are paid once per LB
// elements.  Loads run one group of U loop steps ahead of the arithmetic (two register buffers).
//
// fused = 1 (grid-level S == 1): the workgroup finishes the update itself,
//     A_new = A .* numerator ./ sumB           (Mult:67 / Mult:70, same operation order)
// and publishes its partial sum of A_new (one slot per tile) for the other half-step's denominators.
// fused = 0: partial numerators go to `part` and reduce_kernel finishes.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float div_t(float x, float p) {
#if NMFK_FASTDIV == 2
  return x * __builtin_amdgcn_rcpf(p);      // v_rcp_f32: 1 ulp
#elif NMFK_FASTDIV
  float r = __builtin_amdgcn_rcpf(p);       // 1 ulp
  r = fmaf(fmaf(-p, r, 1.0f), r, r);        // one Newton step: 1/p to ~0.5 ulp
  return x * r;
#else
  return x / p;
#endif
}
__device__ __forceinline__ double div_t(double x, double p) { return x / p; }

// Two-wide vectors: on gfx950 fp32 pairs map to the packed VALU ops (v_pk_fma_f32 ...).  A lone wave issues one
// VALU instruction per 4 cycles, which a packed op fills completely and a scalar fp32 op only half, so the
// inner loop is written on pairs: the two lane elements of a thread (LB = 2) or adjacent signals (LB = 1).
typedef T T2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ T2 fma2(T2 a, T2 b, T2 c) { return __builtin_elementwise_fma(a, b, c); }
// RULE (gfx950 hazard, DESIGN.md "Known hazard"): a broadcast operand of a packed multiply / FMA is written FIRST.  hipcc
// keeps the source order, so the half select of a broadcast from the odd register of a VGPR pair lands on src0
// (op_sel:[1,0,0]), which is safe; on src1 (op_sel:[0,1,0]) the instruction returns wrong low halves in the lanes 48-63
// while another wave on the CU issues 128-bit-operand matrix instructions.  scripts/isa_lint_pk_opsel.py checks the
// generated code of every kernel (tests/test_isa_lint.py).
__device__ __forceinline__ T2 splat2(T v) { return (T2)(v); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 div2(f32x2 x, f32x2 p) {
#if NMFK_FASTDIV
  f32x2 r = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
#if NMFK_FASTDIV != 2
  r = __builtin_elementwise_fma(__builtin_elementwise_fma(-p, r, (f32x2)(1.0f)), r, r);
#endif
  return x * r;
#else
  return x / p;
#endif
}
__device__ __forceinline__ f64x2 div2(f64x2 x, f64x2 p) { return x / p; }

template <int KP, int LB, bool NANS>
__device__ __forceinline__ void step_body(char *arena, const float *__restrict__ X, const NmfkStepArgs *__restrict__ gp,
                                          const NmfkRun *__restrict__ rdp, const int it, double *lds, const int bx) {
  // loop steps per group: the loop-factor rows of TWO groups live in SGPRs (about 100 available)
  constexpr int U = (KP <= 6) ? 4 : (KP <= 12) ? 2 : 1;
  constexpr bool PIPE = KP <= 24;  // wider factors: single buffer, no run-ahead loads
  // Only the pointers are by-value kernel arguments (the compiler must know they are global memory to use
  // scalar loads); the rest of the argument block and the unit descriptor live in device memory and are read
  // field by field where they are needed, so that they do not occupy SGPRs across the main loop.
  struct {
    char *arena;
    const float *X;
    int64_t ld;
    int L, D, S, dchunk, which, wsplit, it, lambda_bits;
  } g;
  g.arena = arena;
  g.X = X;
  g.ld = gp->ld;
  g.L = gp->L;
  g.D = gp->D;
  g.S = gp->S;
  g.dchunk = gp->dchunk;
  g.which = gp->which;
  g.wsplit = gp->wsplit;
  g.it = it;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = g.wsplit;
  const int lpw = (ws > 1) ? 64 : NMFK_TILE;
  const int tile = bx / g.S;  // bx: index of the (lane tile, loop split) pair, see NMFK_GRID
  const int s = bx - tile * g.S;
  if (tile * lpw * LB >= g.L) return;  // the grid is sized for the smallest LB of the launch
  // lane elements of a thread: lpw apart (adjacent ones with one 8-byte X load measured 5-10 % slower)
  const int lbase = tile * lpw * LB + ((ws > 1) ? lane : tid);

  const T *__restrict__ Hcur = NMFK_PTR(const T, g, NMFK_HOFF(*rdp, g.it));
  const T *__restrict__ Hnew = NMFK_PTR(const T, g, NMFK_HOFF(*rdp, g.it + 1));
  const T *__restrict__ Wt = NMFK_PTR(const T, g, rdp->oWt);
  const T *__restrict__ A = g.which == 0 ? Hcur : Wt;  // lane factor
  const T *__restrict__ B = g.which == 0 ? Wt : Hnew;  // loop factor (wave-uniform rows)
  const T *__restrict__ Bold = Hcur;                   // W half-step, missing data: H before this iteration

  // Pairs of adjacent signals (c = 2j, 2j+1) are the packed lanes: the loop-factor row then supplies natural,
  // aligned SGPR pairs (b[2j], b[2j+1]) and nothing has to be broadcast or realigned.  Odd KP: one scalar tail.
  constexpr int NP = KP / 2, NPA = NP > 0 ? NP : 1;
  constexpr bool TAIL = (KP & 1) != 0;
  bool valid[LB];
  int lc[LB];
  // NMFK_EPACK: with two lane elements per thread the packed lanes are the two ELEMENTS instead (signal c of both
  // elements in one register pair, the loop-factor entry b[c] broadcast to both halves): no horizontal add of the
  // two halves of p, no odd-rank tail, and the ratio pair comes out packed.
  constexpr bool EP = (NMFK_EPACK != 0) && LB == 2;
  T2 a2[LB][NPA], acc2[LB][NPA];
  T at[LB], acct[LB];
  T2 ae[EP ? KP : 1], acce[EP ? KP : 1];
#define A_(e, c) (EP ? ae[EP ? (c) : 0][(e)] : (TAIL && (c) == KP - 1) ? at[(e)] : a2[(e)][(c) / 2][(c) & 1])
#define ACC_(e, c) (EP ? acce[EP ? (c) : 0][(e)] : (TAIL && (c) == KP - 1) ? acct[(e)] : acc2[(e)][(c) / 2][(c) & 1])
#pragma unroll
  for (int e = 0; e < LB; ++e) {
    const int l = lbase + e * lpw;
    valid[e] = l < g.L;
    lc[e] = valid[e] ? l : 0;
    at[e] = (T)0;
    acct[e] = (T)0;
#pragma unroll
    for (int c = 0; c < KP; ++c) {
      const T v = valid[e] ? A[c + (int64_t)lc[e] * KP] : (T)1;
      if (EP)
        ae[EP ? c : 0][e] = v;
      else if (TAIL && c == KP - 1)
        at[e] = v;
      else
        a2[e][c / 2][c & 1] = v;
    }
#pragma unroll
    for (int j = 0; j < NPA; ++j) acc2[e][j] = splat2((T)0);
  }
#pragma unroll
  for (int c = 0; c < (EP ? KP : 1); ++c) acce[c] = splat2((T)0);
  int d0 = s * g.dchunk;
  int d1 = min(g.D, d0 + g.dchunk);
  if (ws > 1) {  // the ws waves of the workgroup share the lane elements and split the loop range
    const int q = (d1 - d0 + ws - 1) / ws;
    d0 = min(d0 + wave * q, d1);
    d1 = min(d0 + q, d1);
  }
  d0 = __builtin_amdgcn_readfirstlane(d0);
  d1 = __builtin_amdgcn_readfirstlane(d1);

  // running, wave-uniform row pointers (SGPR pairs); the per-lane part of the X address is a 32-bit offset
  unsigned lofs[LB];
#pragma unroll
  for (int e = 0; e < LB; ++e) lofs[e] = (unsigned)lc[e];
  const float *__restrict__ xnext = g.X + (int64_t)d0 * g.ld;  // row of the next group to LOAD
  const T *__restrict__ bnext = B + (int64_t)d0 * KP;
  const int64_t ld = g.ld;

#if NMFK_XBUF
  // X entries come through BUFFER loads: address = resource base (the group's first row, wave-uniform, SGPRs) +
  // soffset (row within the group, SGPR) + per-lane 32-bit byte offset (VGPR, loop invariant).  All address
  // arithmetic is scalar; with global loads the compiler keeps one 64-bit VGPR base per (row, element) and adds the
  // running offset with a VALU instruction per load (1 of every 5 VALU issue slots at k = 4).
  unsigned lbyte[LB];
#pragma unroll
  for (int e = 0; e < LB; ++e) lbyte[e] = lofs[e] * 4u;
  const int ldb = (int)(ld * 4);  // bytes per row (nmfk_set_X limits the dimensions to 2^27)
#endif
  auto load = [&](T (&bufv)[U][KP], float (&bufx)[U][LB]) __attribute__((always_inline)) {
#if NMFK_XBUF
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)xnext, 0, -1, 0x00020000);
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
#pragma unroll
      for (int c = 0; c < KP; ++c) bufv[uu][c] = bnext[uu * KP + c];
#pragma unroll
      for (int e = 0; e < LB; ++e)
        bufx[uu][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lbyte[e], uu * ldb, 0));
    }
    xnext += U * ld;
#else
    const float *__restrict__ xr = xnext;
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
#pragma unroll
      for (int c = 0; c < KP; ++c) bufv[uu][c] = bnext[uu * KP + c];
#pragma unroll
      for (int e = 0; e < LB; ++e) bufx[uu][e] = xr[lofs[e]];
      xr += ld;
    }
    xnext = xr;
#endif
    bnext += U * KP;
  };
  // imputed value of a missing entry (EM imputation, Mult:72): fl32(W*H) of the previous iteration's result,
  // lambda on the first iteration (Mult:20).  H half-step: that IS p.  W half-step: <w_old, h_old>.
  auto impute = [&](int d, int e, T p) __attribute__((always_inline)) -> T {
    if (g.it == 0) return (T)gp->lambda;
    if (g.which == 0) return (T)(float)p;
    const T *__restrict__ bo = Bold + (int64_t)d * KP;
    T po = (T)0;
#pragma unroll
    for (int c = 0; c < KP; ++c) po = fma_t(A_(e, c), bo[c], po);
    return (T)(float)po;
  };
  auto row = [&](int d, const T *bv, const float *xf) __attribute__((always_inline)) {
    if (EP) {
      // (2 or 4 independent partial sums instead of one chain of KP dependent FMAs: no gain measured)
      T2 pp[1] = {splat2((T)0)};
#pragma unroll
      for (int c = 0; c < KP; ++c) {
#ifdef NMFK_UNSAFE_OPERAND_ORDER  // (tools/hazard/build_hazard_lib.sh: the order that makes hipcc emit the unsafe select)
        pp[0] = fma2(ae[EP ? c : 0], splat2(bv[c]), pp[0]);
#else
        pp[0] = fma2(splat2(bv[c]), ae[EP ? c : 0], pp[0]);
#endif
      }
      T2 p2 = pp[0];
      T2 x2 = {(T)xf[0], (T)xf[LB - 1]};
      if (NANS) {
#pragma unroll
        for (int e = 0; e < LB; ++e) {
          const bool isn = xf[e] != xf[e];
          if (__any(isn)) {
            const T xi = impute(d, e, p2[e]);
            x2[e] = isn ? xi : x2[e];
          }
        }
      }
      const T2 q2 = div2(x2, p2);
#pragma unroll
      for (int c = 0; c < KP; ++c) acce[EP ? c : 0] = fma2(splat2(bv[c]), q2, acce[EP ? c : 0]);
      return;
    }
    T p[LB], q[LB], x[LB];
#pragma unroll
    for (int e = 0; e < LB; ++e) {
      T2 s2 = splat2((T)0);
#pragma unroll
      for (int j = 0; j < NP; ++j) s2 = fma2(a2[e][j], (T2){bv[2 * j], bv[2 * j + 1]}, s2);
      T pe = s2.x + s2.y;
      if (TAIL) pe = fma_t(at[e], bv[KP - 1], pe);
      p[e] = pe;
      x[e] = (T)xf[e];
      if (NANS) {
        const bool isn = xf[e] != xf[e];
        if (__any(isn)) {
          const T xi = impute(d, e, pe);
          x[e] = isn ? xi : x[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e + 1 < LB; e += 2) {  // ratios two at a time (packed Newton step)
      const T2 q2 = div2((T2){x[e], x[e + 1]}, (T2){p[e], p[e + 1]});
      q[e] = q2.x;
      q[e + 1] = q2.y;
    }
    if (LB & 1) q[LB - 1] = div_t(x[LB - 1], p[LB - 1]);
#pragma unroll
    for (int e = 0; e < LB; ++e) {
#pragma unroll
      for (int j = 0; j < NP; ++j) acc2[e][j] = fma2(splat2(q[e]), (T2){bv[2 * j], bv[2 * j + 1]}, acc2[e][j]);
      if (TAIL) acct[e] = fma_t(bv[KP - 1], q[e], acct[e]);
    }
  };
  auto compute = [&](int d, const T (&bufv)[U][KP], const float (&bufx)[U][LB]) __attribute__((always_inline)) {
#pragma unroll
    for (int uu = 0; uu < U; ++uu) row(d + uu, bufv[uu], bufx[uu]);
  };

  int d = d0;
  const int nfull = (d1 - d) / U;
  if (nfull > 0) {
    T v0[U][KP], v1[U][KP];  // the two register buffers: loop-factor rows (SGPRs) ...
    float x0[U][LB], x1[U][LB];  // ... and X entries (VGPRs)
    load(v0, x0);
    if (PIPE) {
      // steady state: two groups per trip, loads one group ahead, buffers alternate without copies
      // (sched_barrier: keep the loads where they are written -- the scheduler would sink them to their uses.
      //  Scalar loads return out of order, so the only wait is lgkmcnt(0); it lands at the top of each
      //  compute phase, one whole phase after the load was issued.)
      for (int pairs = (nfull - 1) >> 1; pairs > 0; --pairs) {
        load(v1, x1);
        __builtin_amdgcn_sched_barrier(0);
        compute(d, v0, x0);
        __builtin_amdgcn_sched_barrier(0);
        load(v0, x0);
        __builtin_amdgcn_sched_barrier(0);
        compute(d + U, v1, x1);
        __builtin_amdgcn_sched_barrier(0);
        d += 2 * U;
      }
      if (((nfull - 1) & 1) != 0) {  // two groups left, the first already loaded
        load(v1, x1);
        compute(d, v0, x0);
        compute(d + U, v1, x1);
        d += 2 * U;
      } else {
        compute(d, v0, x0);
        d += U;
      }
    } else {
      compute(d, v0, x0);
      d += U;
      for (int gi = 1; gi < nfull; ++gi, d += U) {
        load(v0, x0);
        compute(d, v0, x0);
      }
    }
  }
  for (; d < d1; ++d) {  // remainder rows
    const T *__restrict__ b = B + (int64_t)d * KP;
    T bv[KP];
    float xf[LB];
#pragma unroll
    for (int c = 0; c < KP; ++c) bv[c] = b[c];
#pragma unroll
    for (int e = 0; e < LB; ++e) xf[e] = g.X[(int64_t)d * g.ld + lc[e]];
    row(d, bv, xf);
  }

  // wsplit = 4: numerators of waves 1..3 are added to wave 0's in wave order (deterministic)
  T *ldsT = (T *)(lds + 9 * NMFK_MAX_K);  // cross-wave scratch behind den[64] and red[8*64]
  if (ws > 1) {
    if (wave > 0) {
#pragma unroll
      for (int e = 0; e < LB; ++e)
#pragma unroll
        for (int c = 0; c < KP; ++c) ldsT[(((wave - 1) * LB + e) * KP + c) * 64 + lane] = ACC_(e, c);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll 1
      for (int w = 0; w < ws - 1; ++w) {  // not unrolled: one wave's worth of temporaries at a time
#pragma unroll
        for (int e = 0; e < LB; ++e)
#pragma unroll
          for (int c = 0; c < KP; ++c) {
            const T v = ldsT[((w * LB + e) * KP + c) * 64 + lane];
            if (EP)
              acce[EP ? c : 0][e] += v;
            else if (TAIL && c == KP - 1)
              acct[e] += v;
            else
              acc2[e][c / 2][c & 1] += v;
          }
      }
    }
    __syncthreads();
  }
  const bool owner = (ws == 1) || (wave == 0);

  if (!gp->fused) {
    if (owner) {
#pragma unroll
      for (int e = 0; e < LB; ++e)
        if (valid[e]) {
          T *__restrict__ part = NMFK_PTR(T, g, rdp->opart) + ((int64_t)s * g.L + lc[e]) * KP;
#pragma unroll
          for (int c = 0; c < KP; ++c) part[c] = ACC_(e, c);
        }
    }
    return;
  }

  // fused finish.  denominators: sum of the other factor's per-tile partial sums (fp64, fixed order)
  const double *sumB = NMFK_PTR(const double, g, g.which == 0 ? rdp->osumW : rdp->osumH);
  const int PB = g.which == 0 ? gp->PW : gp->PH;
  double *den = lds;
  if (tid < KP) den[tid] = nmfk_slot_sum<4>(sumB, KP, PB, tid);
  __syncthreads();
  T *__restrict__ Anew = g.which == 0 ? NMFK_PTR(T, g, NMFK_HOFF(*rdp, g.it + 1)) : NMFK_PTR(T, g, rdp->oWt);
  const int k = rdp->k;
  // per-workgroup partial sums of A_new -> slot `tile` of the sum table of this factor
  double *sumA = NMFK_PTR(double, g, g.which == 0 ? rdp->osumH : rdp->osumW) + (int64_t)tile * KP;
  double *red = den + NMFK_MAX_K;  // [8][KP]
#pragma unroll
  for (int c = 0; c < KP; ++c) {
    T vs = (T)0;
    if (owner) {
#pragma unroll
      for (int e = 0; e < LB; ++e) {
        T v = A_(e, c) * ACC_(e, c) / (T)den[c];
        if (c >= k || !valid[e]) v = (T)0;
        if (valid[e]) Anew[c + (int64_t)lc[e] * KP] = v;
        vs += v;
      }
    }
    const double v = wave_sum((double)vs);
    if (lane == 0) red[wave * KP + c] = v;
  }
  __syncthreads();
  if (tid < KP) {
    const double t = (ws > 1) ? red[tid] : ((red[tid] + red[KP + tid]) + (red[2 * KP + tid] + red[3 * KP + tid]));
    sumA[tid] = t;
  }
}

// One kernel per rank width KP: the register allocation (hence occupancy) of a kernel is the maximum over
// everything it can dispatch to, and a switch over ranks inside one kernel inflates it well beyond the widest
// case.  Units of equal rank are contiguous (sorted by k), so a launch covers the unit range [u0, u0 + gridDim.y).
// LDS (dynamic, sized by the launcher): den[64], red[8*64], then the cross-wave scratch of (ws-1)*LB*KP*64 elements
// of T (wsplit > 1 only).
// min waves per SIMD requested from the register allocator (2nd __launch_bounds__ argument = waves per EU)
#ifndef NMFK_MINWAVES
#define NMFK_MINWAVES(KP) ((KP) <= 16 ? 4 : 1)
#endif
template <bool NANS, int KP>
__global__ __launch_bounds__(2 * NMFK_TILE, NMFK_MINWAVES(KP)) void step_kernel(char *arena, const float *__restrict__ X,
                                                         const NmfkRun *__restrict__ runs,
                                                         const NmfkState *__restrict__ state,
                                                         const NmfkStepArgs *__restrict__ gp, int it, int u0, int uf) {
  constexpr int LB = NMFK_LB_OF(KP);
  extern __shared__ double lds[];
  const int u = u0 + (uf ? blockIdx.x : blockIdx.y);
  if (!gp->force && !state[u].active) return;
  step_body<KP, LB, NANS>(arena, X, gp, runs + u, it, lds, uf ? blockIdx.y : blockIdx.x);
}

#if !defined(NMFK_IS_F32) || NMFK_WITH_MERGED_F32  // (fp32: see NMFK_WITH_MERGED_F32)
// One launch for units of DIFFERENT ranks (all <= 16, so that they share the lane tile 64/256 * NMFK_LB): used when a
// sweep has so few restarts per rank that per-rank launches leave the loop launch-bound (strong scaling over many
// GPUs).  The register allocation is that of the widest case, which is irrelevant when the chip is not full anyway.
#define NMFK_MULTI_CASE(KP) step_body<KP, NMFK_MULTI_LB, NANS>(arena, X, gp, runs + u, it, lds, bx)
template <bool NANS>
__global__ __launch_bounds__(2 * NMFK_TILE, NMFK_MULTI_MINWAVES) void step_kernel_multi(char *arena, const float *__restrict__ X,
                                                                      const NmfkRun *__restrict__ runs,
                                                                      const NmfkState *__restrict__ state,
                                                                      const NmfkStepArgs *__restrict__ gp, int it, int u0,
                                                                      int uf) {
  extern __shared__ double lds[];
  const int u = u0 + (uf ? blockIdx.x : blockIdx.y);
  const int bx = uf ? blockIdx.y : blockIdx.x;
  if (!gp->force && !state[u].active) return;
  switch (runs[u].kp) {
    case 1: NMFK_MULTI_CASE(1); break;   case 2: NMFK_MULTI_CASE(2); break;   case 3: NMFK_MULTI_CASE(3); break;
    case 4: NMFK_MULTI_CASE(4); break;   case 5: NMFK_MULTI_CASE(5); break;   case 6: NMFK_MULTI_CASE(6); break;
    case 7: NMFK_MULTI_CASE(7); break;   case 8: NMFK_MULTI_CASE(8); break;
#if NMFK_MULTI_MAXK >= 12
    case 9: NMFK_MULTI_CASE(9); break;   case 10: NMFK_MULTI_CASE(10); break; case 11: NMFK_MULTI_CASE(11); break;
    case 12: NMFK_MULTI_CASE(12); break;
#endif
#if NMFK_MULTI_MAXK >= 14
    case 13: NMFK_MULTI_CASE(13); break; case 14: NMFK_MULTI_CASE(14); break;
#endif
#if NMFK_MULTI_MAXK >= 16
    case 15: NMFK_MULTI_CASE(15); break; case 16: NMFK_MULTI_CASE(16); break;
#endif
    default: break;
  }
}
#endif

#ifdef NMFK_IS_F32
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte global access
#endif

#ifdef NMFK_IS_F32
// ------------------------------------------------------------------------------------------------------
// All-MFMA half-step for wide ranks (16 < k <= 64; kp is a multiple of 4, padding rows of the factors are zero).
// The VALU kernel needs 4k VGPRs per thread for a and acc, which leaves 1-3 waves/SIMD at k >= 32; here the
// numerators live in MFMA accumulator tiles: a wave owns NT tiles of 16 lane elements and NB = ceil(kp/16) blocks
// of 16 signals, i.e. NT*NB accumulators of 4 VGPRs.  Per 16 loop steps and tile: KQ = kp/4 MFMAs for
// P = B'A (16 x 16), 16 reciprocals on the VALU, 4*NB MFMAs for N += B Q.
// ------------------------------------------------------------------------------------------------------
template <int KQ, int NB, int NT>
__global__ __launch_bounds__(2 * NMFK_TILE) void mfma_wide_kernel(char *arena, const float *__restrict__ X,
                                                                 const NmfkRun *__restrict__ runs,
                                                                 const NmfkState *__restrict__ state,
                                                                 const NmfkStepArgs *__restrict__ gp, int it, int u0,
                                                                 int uf) {
  extern __shared__ double lds[];  // den[64], red[8*64], then max(staging, cross-wave scratch)
  constexpr int KP = 4 * KQ, RS = KP + 4;  // staged row stride (floats): 16-byte aligned, off the 32-bank period
  int u = u0 + (uf ? blockIdx.x : blockIdx.y);
  int bx = uf ? blockIdx.y : blockIdx.x;
#if NMFK_XCD_REMAP
  // Large X (tile slice of an XCD >> its 4 MB L2): workgroups are dealt to the XCDs round-robin in linear order, so
  // make the restarts of ONE lane tile consecutive workgroups of ONE XCD -- they run together and share the tile's X
  // rows through that L2 instead of fetching them once per restart.  linear id -> (xcd, slot); slot -> (tile group,
  // unit); tile = 8 * group + xcd.
  if (!uf && (gridDim.x & 7) == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, nu = gridDim.y;
    const int xcd = lin & 7, slot = lin >> 3;
    u = u0 + slot % nu;
    bx = (slot / nu) * 8 + xcd;
  }
#endif
  if (!gp->force && !state[u].active) return;
  const NmfkRun *__restrict__ rdp = runs + u;
  const int k = rdp->k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int which = gp->which, ws = gp->wsplit, S = gp->S, L = gp->L, D = gp->D;
  const int nwaves = blockDim.x >> 6;
  const int lpw = 16 * NT * (ws > 1 ? 1 : nwaves);
  const int tile = bx / S, s = bx - tile * S;
  const int l0 = tile * lpw + (ws > 1 ? 0 : wave * 16 * NT);

  const float *__restrict__ Hcur = (const float *)(arena + NMFK_HOFF(*rdp, it));
  const float *__restrict__ Hnew = (const float *)(arena + NMFK_HOFF(*rdp, it + 1));
  const float *__restrict__ Wt = (const float *)(arena + rdp->oWt);
  const float *__restrict__ A = which == 0 ? Hcur : Wt;  // lane factor
  const float *__restrict__ B = which == 0 ? Wt : Hnew;  // loop factor

  int d0 = s * gp->dchunk;
  int d1 = min(D, d0 + gp->dchunk);
  if (ws > 1) {
    const int q = (((d1 - d0 + ws - 1) / ws) + 15) & ~15;  // equal shares of the range per wave, in whole chunks
    d0 = min(d0 + wave * q, d1);
    d1 = min(d0 + q, d1);
  }
  d0 = __builtin_amdgcn_readfirstlane(d0);
  d1 = __builtin_amdgcn_readfirstlane(d1);

  // first product: contraction index (MFMA step sq, k-lane g) <-> signal c = KQ*g + sq
  float afrag[NT][KQ];
  int lt[NT];
  bool lv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int l = l0 + 16 * t + c16;
    lv[t] = l < L;
    lt[t] = lv[t] ? l : 0;
#pragma unroll
    for (int sq = 0; sq < KQ; ++sq) afrag[t][sq] = lv[t] ? A[KQ * g + sq + (int64_t)lt[t] * KP] : 0.0f;
  }
  f32x4_t acc[NT][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const float *__restrict__ Xa = gp->Xalt;  // element (l, d) at d + l*D
  float *scratch = (float *)(lds + 9 * NMFK_MAX_K);
  float *stage = scratch + wave * (16 * RS);
  const int nch = (d1 - d0 + 15) >> 4;
  const float *xbase[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xbase[t] = Xa + (int64_t)lt[t] * D + 4 * g;
  // this lane's pieces (16 bytes each) of a 16 x KP chunk: piece pi = lane + 64q, row (4 pi)/KP, column (4 pi)%KP
  int wofs[NB];
  bool wv[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int pi = lane + 64 * q;
    wv[q] = pi < 4 * KP;
    wofs[q] = ((4 * pi) / KP) * RS + (4 * pi) % KP;
  }
  const int pofs = c16 * RS + KQ * g;
  const int nofs = 4 * g * RS + c16;

  auto load = [&](int dch, f32x4_t (&xv)[NT], f32x4_t (&bv)[NB]) __attribute__((always_inline)) {
    const int dx = (dch + 16 <= D) ? dch : (min(dch + 4 * g, D - 4) - 4 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t) xv[t] = *(const f32x4_u *)(xbase[t] + dx);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      bv[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (wv[q]) bv[q] = *(const f32x4_u *)(B + (int64_t)dch * KP + 4 * (lane + 64 * q));
    }
  };
  auto chunk = [&](int dch, const f32x4_t (&xcur)[NT], auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    float bP[KQ];
    {
      const float *pr = stage + (FULL ? pofs : min(c16, d1 - 1 - dch) * RS + KQ * g);
#pragma unroll
      for (int sq = 0; sq < KQ; ++sq) bP[sq] = pr[sq];
    }
    bool rv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[r] = FULL || (dch + 4 * g + r < d1);
    f32x4_t p[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sq = 0; sq < KQ; ++sq)
#pragma unroll
      for (int t = 0; t < NT; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[sq], afrag[t][sq], p[t], 0, 0, 0);
    f32x4_t q[NT];
    if (FULL) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) q[t][r] = div_t(xcur[t][r], p[t][r]);
    } else {
      const int shift = (dch + 4 * g) - min(dch + 4 * g, D - 4);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = r + shift;
          const float xx = rr <= 0 ? xcur[t][0] : rr == 1 ? xcur[t][1] : rr == 2 ? xcur[t][2] : xcur[t][3];
          q[t][r] = rv[r] ? div_t(xx, p[t][r]) : 0.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float bN = (rv[r] && 16 * nb + c16 < KP) ? stage[nofs + RS * r + 16 * nb] : 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bN, q[t][r], acc[t][nb], 0, 0, 0);
      }
  };
  auto step = [&](int ci, const f32x4_t (&xc)[NT], const f32x4_t (&bc)[NB], f32x4_t (&xn)[NT], f32x4_t (&bn)[NB])
                  __attribute__((always_inline)) {
    const int dch = d0 + 16 * ci;
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (wv[q]) *(f32x4_t *)(stage + wofs[q]) = bc[q];
    if (ci + 1 < nch) load(dch + 16, xn, bn);
    __builtin_amdgcn_wave_barrier();
    if (dch + 16 <= d1 && dch + 16 <= D)
      chunk(dch, xc, std::true_type());
    else
      chunk(dch, xc, std::false_type());
    __builtin_amdgcn_wave_barrier();
  };
  {
    f32x4_t x0[NT], x1[NT], b0[NB], b1[NB];
    if (nch > 0) load(d0, x0, b0);
    for (int ci = 0; ci < nch; ci += 2) {
      step(ci, x0, b0, x1, b1);
      if (ci + 1 < nch) step(ci + 1, x1, b1, x0, b0);
    }
  }
  // acc[t][nb][r] = numerator of signal c = 16nb + 4g + r at lane element l0 + 16t + c16

  if (ws > 1) {  // add the waves' numerators in wave order (the scratch overlays the staging buffers)
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) scratch[((((wave - 1) * NT + t) * NB + nb) * 4 + r) * 64 + lane] = acc[t][nb][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll 1
      for (int w = 0; w < ws - 1; ++w)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][nb][r] += scratch[(((w * NT + t) * NB + nb) * 4 + r) * 64 + lane];
    }
    __syncthreads();
  }
  const bool owner = (ws == 1) || (wave == 0);

  if (!gp->fused) {
    if (owner) {
      float *__restrict__ part = (float *)(arena + rdp->opart);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (lv[t]) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int c = 16 * nb + 4 * g + r;
              if (c < KP) part[((int64_t)s * L + lt[t]) * KP + c] = acc[t][nb][r];
            }
        }
    }
    return;
  }

  const double *sumB = (const double *)(arena + (which == 0 ? rdp->osumW : rdp->osumH));
  const int PB = which == 0 ? gp->PW : gp->PH;
  double *den = lds;
  if (tid < KP) den[tid] = nmfk_slot_sum<4>(sumB, KP, PB, tid);
  __syncthreads();
  float *__restrict__ Anew = which == 0 ? (float *)(arena + NMFK_HOFF(*rdp, it + 1)) : (float *)(arena + rdp->oWt);
  double *sumA = (double *)(arena + (which == 0 ? rdp->osumH : rdp->osumW)) + (int64_t)tile * KP;
  double *red = den + NMFK_MAX_K;  // [8][KP]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * nb + 4 * g + r;
      float vs = 0.f;
      if (owner && c < KP) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (lv[t]) {
            float v = 0.f;
            if (c < k) v = A[c + (int64_t)lt[t] * KP] * acc[t][nb][r] / (float)den[c];  // Mult:67 / Mult:70 order
            Anew[c + (int64_t)lt[t] * KP] = v;
            vs += v;
          }
      }
      double v = (double)vs;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (c16 == 0 && c < KP) red[wave * KP + c] = v;
    }
  __syncthreads();
  if (tid < KP) {
    double t = red[tid];
    if (ws == 1)
      for (int w = 1; w < nwaves; ++w) t += red[w * KP + tid];
    sumA[tid] = (tid < k) ? t : 0.0;
  }
}


// Monitored objective (Mult:74) for ranks above 16 on the matrix pipe: the W half-step's first product only,
// P = H' W' tile by tile, then sum((x - p)^2 * weight^2) in fp64.  fp32, no missing data, scalar weight; lanes = rows of X.
template <int KQ>
__global__ __launch_bounds__(NMFK_TILE) void mfma_sse_kernel(NmfkSseArgs g, int u0) {
  extern __shared__ double lds[];  // [8] block-sum scratch, then one 16 x KP staging buffer per wave
  constexpr int KP = 4 * KQ, RS = KP + 4, NT = 4, NB = (KP + 15) / 16;
  const int u = u0 + blockIdx.y;
  const NmfkState st = g.state[u];
  if (!g.force && !st.active) return;
  const NmfkRun *__restrict__ rdp = g.runs + u;
  const int sel = g.hsel >= 0 ? g.hsel : ((st.active ? g.total_iters : st.iters) & 1);
  const float *__restrict__ B = (const float *)(g.arena + NMFK_HOFF(*rdp, sel));
  const float *__restrict__ A = (const float *)(g.arena + rdp->oWt);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, c16 = lane & 15;
  const int L = g.n, D = g.m;
  const int l0 = blockIdx.x * NMFK_TILE + wave * 64;
  float afrag[NT][KQ];
  int lt[NT];
  bool lv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int l = l0 + 16 * t + c16;
    lv[t] = l < L;
    lt[t] = lv[t] ? l : 0;
#pragma unroll
    for (int sq = 0; sq < KQ; ++sq) afrag[t][sq] = lv[t] ? A[KQ * gq + sq + (int64_t)lt[t] * KP] : 0.0f;
  }
  float *stage = (float *)(lds + 8) + wave * (16 * RS);
  const float *xbase[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xbase[t] = g.Xr + (int64_t)lt[t] * D + 4 * gq;
  int wofs[NB];
  bool wv[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int pi = lane + 64 * q;
    wv[q] = pi < 4 * KP;
    wofs[q] = ((4 * pi) / KP) * RS + (4 * pi) % KP;
  }
  const float wgt = (float)g.weight;
  double ss = 0.0;
  const int nch = (D + 15) >> 4;
  auto load = [&](int dch, f32x4_t (&xv)[NT], f32x4_t (&bv)[NB]) __attribute__((always_inline)) {
    const int dx = (dch + 16 <= D) ? dch : (min(dch + 4 * gq, D - 4) - 4 * gq);
#pragma unroll
    for (int t = 0; t < NT; ++t) xv[t] = *(const f32x4_u *)(xbase[t] + dx);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      bv[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (wv[q]) bv[q] = *(const f32x4_u *)(B + (int64_t)dch * KP + 4 * (lane + 64 * q));
    }
  };
  auto step = [&](int ci, const f32x4_t (&xc)[NT], const f32x4_t (&bc)[NB], f32x4_t (&xn)[NT], f32x4_t (&bn)[NB])
                  __attribute__((always_inline)) {
    const int dch = 16 * ci;
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (wv[q]) *(f32x4_t *)(stage + wofs[q]) = bc[q];
    if (ci + 1 < nch) load(dch + 16, xn, bn);
    __builtin_amdgcn_wave_barrier();
    const bool full = dch + 16 <= D;
    float bP[KQ];
    {
      const float *pr = stage + (full ? c16 : min(c16, D - 1 - dch)) * RS + KQ * gq;
#pragma unroll
      for (int sq = 0; sq < KQ; ++sq) bP[sq] = pr[sq];
    }
    f32x4_t p[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sq = 0; sq < KQ; ++sq)
#pragma unroll
      for (int t = 0; t < NT; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[sq], afrag[t][sq], p[t], 0, 0, 0);
    const int shift = full ? 0 : (dch + 4 * gq) - min(dch + 4 * gq, D - 4);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = r + shift;
        const float xx = rr <= 0 ? xc[t][0] : rr == 1 ? xc[t][1] : rr == 2 ? xc[t][2] : xc[t][3];
        const float e = (xx - p[t][r]) * wgt;
        const bool use = lv[t] && (dch + 4 * gq + r < D);
        ss += use ? (double)e * (double)e : 0.0;
      }
    __builtin_amdgcn_wave_barrier();
  };
  {
    f32x4_t x0[NT], x1[NT], b0[NB], b1[NB];
    load(0, x0, b0);
    for (int ci = 0; ci < nch; ci += 2) {
      step(ci, x0, b0, x1, b1);
      if (ci + 1 < nch) step(ci + 1, x1, b1, x0, b0);
    }
  }
  ss = block_sum(ss, lds);
  if (tid == 0) ((double *)(g.arena + rdp->ossepart))[blockIdx.x] = ss;
}
#endif

// ------------------------------------------------------------------------------------------------------
// Sparse X (BASELINE configs[3]: zeros stay zeros).  In the reference a zero becomes lambda = 1e-32 (Mult:17-18), so
// its ratio X/(W*H) is ~1e-32 and only the stored non-zeros contribute to the numerators: the gather form below is
// the reference arithmetic to < 1e-30.  Per output lane element l (a column of H via CSC, a row of W via CSR) and
// non-zero (d, x) of it:      p = <a_l, b_d>;  q = x / p;  acc_l += q * b_d ;   A_new = A .* acc ./ sumB  (fused finish)
// The walk is bound by the gathers of b_d (one row of the other factor, kp*4 bytes, per non-zero), so they are made
// as cheap as the hardware allows: a lane element is owned by a GROUP of LPR adjacent lanes, lane `sub` holding the
// signals [4 sub, 4 sub + 4) of a, acc and of every gathered row -- ONE 16-byte load per lane fetches a whole row
// per group, 64 / LPR rows per wave instruction (a thread per lane element reading its row with kp/4 loads touched
// 64 different cache lines with every instruction: 6 % of HBM speed in round 1).  The non-zeros are 8-byte
// (index, value) records; the lanes of a group read consecutive records with one load and hand them round
// (ds_bpermute), the partial dot products are added across the group by DPP butterflies (every lane of the group
// gets the same bits).  A workgroup still owns NMFK_TILE lane elements = one slot of the sum table and walks them in
// LPR passes of NMFK_TILE / LPR groups.
// ------------------------------------------------------------------------------------------------------
typedef T sp_vec4 __attribute__((ext_vector_type(4)));
typedef T sp_vec4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));
template <int CTRL>
__device__ __forceinline__ float sp_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double sp_dpp(double v) {
  const int2 w = __builtin_bit_cast(int2, v);
  int2 r;
  r.x = __builtin_amdgcn_update_dpp(0, w.x, CTRL, 0xf, 0xf, false);
  r.y = __builtin_amdgcn_update_dpp(0, w.y, CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, r);
}
// sum over the LPR lanes of a group, the same bits in every lane (quad_perm xor 1, xor 2, row_half_mirror, row_mirror)
template <int LPR>
__device__ __forceinline__ T sp_group_sum(T s) {
  if (LPR >= 2) s += sp_dpp<0xB1>(s);
  if (LPR >= 4) s += sp_dpp<0x4E>(s);
  if (LPR >= 8) s += sp_dpp<0x141>(s);
  if (LPR >= 16) s += sp_dpp<0x140>(s);
  return s;
}

// ... of N independent values, level by level: a DPP operand needs two wait states behind the VALU write of its
// source, which the other values' additions fill (one value at a time leaves an s_nop behind every addition)
template <int LPR, int N>
__device__ __forceinline__ void sp_group_sum_n(T (&s)[N]) {
#pragma unroll
  for (int t = 0; t < N; ++t)
    if (LPR >= 2) s[t] += sp_dpp<0xB1>(s[t]);
#pragma unroll
  for (int t = 0; t < N; ++t)
    if (LPR >= 4) s[t] += sp_dpp<0x4E>(s[t]);
#pragma unroll
  for (int t = 0; t < N; ++t)
    if (LPR >= 8) s[t] += sp_dpp<0x141>(s[t]);
#pragma unroll
  for (int t = 0; t < N; ++t)
    if (LPR >= 16) s[t] += sp_dpp<0x140>(s[t]);
}

// the lanes' view of the factor rows and the walk over a lane element's non-zeros, shared by the half-step and the
// objective.  f(t, live, x, p, b): record t of the chunk, `live` false past the end of the range (x and p are then
// those of some other record: the caller drops them), p = <a, b_d> (the same bits in every lane of the group),
// b = this lane's four signals of the gathered row.
// The walk is bound by VALU issue (round 3: 31 instructions per record slot, 7 of them selects), so the inner loop carries
// nothing it does not need:
//  * a lane's four signals start at sig0 = min(4 sub, kp - 4): the last lane of a ragged row (kp not a multiple of 4)
//    re-reads signals of its neighbour instead of reading into the next row, and `own` says which of its four are its
//    own -- a is zero elsewhere, so the dot product needs no mask on the gathered values and nothing outside a row is read
//    (LPR = 1 with kp < 4 is the exception: one lane, rows shorter than its load; the values are masked there);
//  * the row offset d * kp is computed once per fetched record, before the records are handed round;
//  * records past the end of a range are clamped to the array, so their row index is valid without a select.
template <int LPR>
struct SpWalk {
  static constexpr int RL = LPR >= 4 ? 1 : 4 / LPR;  // records a lane fetches per block
  static constexpr int NB = LPR * RL;                // records per block of a group (4, 4, 4, 8, 16)
  static constexpr int CH = NB < 8 ? NB : 8;         // gathers in flight per lane
  int kp, sub, sig0;
  bool on, full, ragged, own[4];
  __device__ __forceinline__ void init(int kp_, int tid) {
    kp = kp_;
    sub = tid % LPR;
    const int first = 4 * sub;
    on = first < kp;  // this lane holds signals of the factor rows
    sig0 = (LPR == 1 || !on) ? 0 : min(first, kp - 4);  // (lanes beyond the row re-read its first signals: no traffic)
#pragma unroll
    for (int e = 0; e < 4; ++e) own[e] = on && sig0 + e >= first && sig0 + e < kp;
    full = own[0] && own[3];
    ragged = LPR == 1 && kp < 4;  // (wave-uniform)
  }
  // Nothing may USE a gathered value between the loads of a chunk: a select or a branch right behind a load makes
  // the compiler wait for vmcnt(0) on the spot and the gathers run one after the other.
  // ofs = row * kp (elements; the sparse path keeps a factor below 2^31 elements -- checked on the host)
  __device__ __forceinline__ sp_vec4 row4(const T *__restrict__ F, int ofs) const {
    return *(const sp_vec4u *)(F + (uint32_t)(ofs + sig0));
  }
  __device__ __forceinline__ sp_vec4 keep(sp_vec4 v) const {
    if (LPR == 1 && ragged) {
#pragma unroll
      for (int e = 1; e < 4; ++e) v[e] = own[e] ? v[e] : (T)0;
    }
    return v;
  }
  __device__ __forceinline__ sp_vec4 mask_own(sp_vec4 a) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = own[e] ? a[e] : (T)0;
    return a;
  }
  __device__ __forceinline__ sp_vec4 lane_row(const T *__restrict__ F, int r) const { return mask_own(row4(F, r * kp)); }
  // <a, b> over this lane's four signals, as two packed operations and an add
  __device__ __forceinline__ T dot4(const sp_vec4 &a, const sp_vec4 &b) const {
    T2 t = T2{a[0], a[1]} * T2{b[0], b[1]};
    t = fma2(T2{a[2], a[3]}, T2{b[2], b[3]}, t);
    return t.x + t.y;
  }
  // acc += q * b
  __device__ __forceinline__ void axpy4(sp_vec4 &acc, T q, const sp_vec4 &b) const {
    const T2 lo = fma2(splat2(q), T2{b[0], b[1]}, T2{acc[0], acc[1]}), hi = fma2(splat2(q), T2{b[2], b[3]}, T2{acc[2], acc[3]});
    acc = sp_vec4{lo.x, lo.y, hi.x, hi.y};
  }
  template <class Fn>
  __device__ __forceinline__ void walk(const NmfkSparseArgs &g, const T *__restrict__ B, const sp_vec4 a, int p0, int p1,
                                       Fn &&f) const {
    const T *__restrict__ Bs = B + sig0;
    walk_rows<CH>(g, kp, 0, [&](int ofs) __attribute__((always_inline)) { return *(const sp_vec4u *)(Bs + (uint32_t)ofs); }, a, p0, p1, f);
  }
  // rowfn(ofs): this lane's four signals of the gathered factor's row d, ofs = (d - d0) * mul (global memory above; the
  // staged block in LDS for the blocked W half-step)
  // CHW: gathers in flight per lane (8 from global memory; LDS answers in ~100 cycles, 4 are plenty and leave registers)
  template <int CHW, class RowFn, class Fn>
  __device__ __forceinline__ void walk_rows(const NmfkSparseArgs &g, int mul, int d0, RowFn &&rowfn, const sp_vec4 a, int p0,
                                            int p1, Fn &&f) const {
    constexpr int CH = NB < CHW ? NB : CHW;
    // a block's records: lane `sub` of the group fetches RL consecutive ones (clamped to the array), one block ahead
    int2 rec[RL], nxt[RL];
#pragma unroll
    for (int j = 0; j < RL; ++j) nxt[j] = g.rec[min(p0 + sub * RL + j, g.nrec - 1)];
    for (int pp = p0; pp < p1; pp += NB) {
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        rec[j] = nxt[j];
        rec[j].x = (rec[j].x - d0) * mul;
        nxt[j] = g.rec[min(pp + NB + sub * RL + j, g.nrec - 1)];
      }
#pragma unroll
      for (int c0 = 0; c0 < NB; c0 += CH) {
        sp_vec4 b[CH];
        T x[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) {  // all gathers of the chunk first: they overlap
          const int ti = c0 + t;
          int ofs = rec[ti % RL].x, xb = rec[ti % RL].y;
          if (LPR > 1) {
            ofs = __shfl(ofs, ti / RL, LPR);
            xb = __shfl(xb, ti / RL, LPR);
          }
          x[t] = (T)__builtin_bit_cast(float, xb);
          b[t] = rowfn(ofs);
        }
        T pr[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          b[t] = keep(b[t]);
          pr[t] = dot4(a, b[t]);
        }
        sp_group_sum_n<LPR, CH>(pr);
#pragma unroll
        for (int t = 0; t < CH; ++t) f(t, pp + c0 + t < p1, x[t], pr[t], b[t]);
      }
    }
  }
};

template <int LPR>
__global__ __launch_bounds__(NMFK_TILE) void sp_step_kernel(NmfkSparseArgs g, int u0, int cnt) {
  constexpr int GPW = NMFK_TILE / LPR;  // groups (lane elements in flight) per workgroup
  __shared__ double lds[5 * NMFK_MAX_K];
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.split) {
    // H half-step: the gathered factor is W (up to 12.8 MB per unit, beyond one XCD's 4 MB of L2), but the columns'
    // non-zeros are sorted by row, so the workgroups of a unit sweep W top-down together and share a moving window
    // of it -- if they sit behind the SAME L2.  Workgroups go round-robin over the 8 XCDs in linear-id order:
    // linear id -> (xcd, slot), slot -> (group of 8 units, workgroup of the unit); unit = 8 * group + xcd.
    // (measured: L2 hit rate of the gathers 19 % with the units spread over the XCDs)
    const int lin = by * gridDim.x + bx, xcd = lin & 7, slot = lin >> 3;
    by = (slot / gridDim.x) * 8 + xcd;
    bx = slot - (slot / gridDim.x) * gridDim.x;
    if (by >= cnt) return;  // (the launcher rounds the unit dimension up to a multiple of 8)
  }
  const int u = u0 + by;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  const int kp = rd.kp, k = rd.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / LPR;
  const int npz = g.split ? LPR : 1, ppw = LPR / npz;  // workgroups per tile, passes per workgroup
  const int tile = bx / npz, pz = bx - tile * npz, slot = bx;
  const T *__restrict__ Hcur = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it));
  const T *__restrict__ Hnew = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it + 1));
  const T *__restrict__ Wt = NMFK_PTR(const T, g, rd.oWt);
  const T *__restrict__ A = g.which == 0 ? Hcur : Wt;
  const T *__restrict__ B = g.which == 0 ? Wt : Hnew;
  T *__restrict__ Anew = g.which == 0 ? NMFK_PTR(T, g, NMFK_HOFF(rd, g.it + 1)) : NMFK_PTR(T, g, rd.oWt);

  // denominators: the other factor's sum table (hundreds of slots).  All threads share the work -- thread (j, c) adds
  // the slots j, j + TPS, ... of signal c, then the TPS partial sums are added in order: a fixed order, so reproducible
  double *den = lds, *red = den + NMFK_MAX_K;  // red: [256] here, [4][4 * LPR] at the end
  {
    const double *sumB = NMFK_PTR(const double, g, g.which == 0 ? rd.osumW : rd.osumH);
    const int PB = g.which == 0 ? rd.nsW : rd.nsH;  // (the slots behind them are zero)
    const int kq = kp <= 4 ? 4 : kp <= 8 ? 8 : kp <= 16 ? 16 : kp <= 32 ? 32 : 64, tps = NMFK_TILE / kq;
    const int c = tid % kq, j = tid / kq;
    double sd = 0;
    if (c < kp) {
#pragma unroll 4
      for (int q = j; q < PB; q += tps) sd += sumB[q * kp + c];
    }
    red[tid] = sd;
    __syncthreads();
    if (tid < kp) {
      sd = 0;
      for (int jj = 0; jj < tps; ++jj) sd += red[jj * kq + tid];
      den[tid] = sd;
    }
  }
  __syncthreads();

  SpWalk<LPR> w;
  w.init(kp, tid);
  double vs[4] = {0, 0, 0, 0};  // this lane's share of the sums of the new factor's signals 4 sub .. 4 sub + 3
#pragma unroll 1
  for (int pass = pz * ppw; pass < (pz + 1) * ppw; ++pass) {
    const int l = tile * NMFK_TILE + pass * GPW + grp;
    const bool valid = l < g.L;
    const int lc = valid ? l : 0;
    const sp_vec4 a = w.lane_row(A, lc);
    sp_vec4 acc = (sp_vec4)((T)0);
    const int p0 = valid ? g.ptr[lc] : 0, p1 = valid ? g.ptr[lc + 1] : 0;
    w.walk(g, B, a, p0, p1, [&](int, bool live, T x, T pr, const sp_vec4 &b) __attribute__((always_inline)) {
      w.axpy4(acc, live ? div_t(x, pr) : (T)0, b);
    });
    // fused finish of the pass (Mult:67 / Mult:70 order)
    if (w.on) {
      sp_vec4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = w.sig0 + e;
        v[e] = (w.own[e] && c < k && valid) ? a[e] * acc[e] / (T)den[c < kp ? c : 0] : (T)0;
        vs[e] += (double)v[e];
      }
      if (valid) {
        T *dst = Anew + (int64_t)lc * kp + w.sig0;
        if (w.full) {
          *(sp_vec4u *)dst = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (w.own[e]) dst[e] = v[e];
        }
      }
    }
  }
  // sums of the new factor over the workgroup's lane elements -> slot `tile`: lanes of equal `sub` across the groups
  // of a wave (xor butterflies over the group index), then the four waves in order
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double v = vs[e];
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane < LPR && w.own[e]) red[wave * 4 * LPR + w.sig0 + e] = v;
  }
  __syncthreads();
  if (tid < kp && slot < (g.which == 0 ? g.PH : g.PW)) {
    double *sumA = NMFK_PTR(double, g, g.which == 0 ? rd.osumH : rd.osumW) + (int64_t)slot * kp;
    sumA[tid] = (red[tid] + red[4 * LPR + tid]) + (red[2 * 4 * LPR + tid] + red[3 * 4 * LPR + tid]);
  }
}

// ------------------------------------------------------------------------------------------------------
// Blocked form of the sparse half-steps (round 3), ranks up to 32.  The gather form above fetches one row of
// the other factor (kp * 4 B) per non-zero through L2 -> L1 -- 105 GB per iteration of BASELINE configs[3], 3.5 x the
// algorithmic bytes -- and spends a group of LPR lanes and ~22 instructions on every non-zero.  Here a workgroup of 1024
// threads owns 1024 lane elements, ONE PER LANE with all its signals in registers (the row of its own factor and the
// numerators), stages the gathered factor through LDS a block of granules at a time (1024 rows of 32 signals fit) and
// serves the non-zeros of its lane elements in that block from LDS: a wave instruction now works on 64 non-zeros instead of
// 8, and a staged row is used ~5 times (0.5 % fill x 1024 lane elements) instead of being fetched ~5 times.
// The non-zeros come as sliced ELL (NmfkSparseArgs::ell): slot row t of a (slice, granule) run is one coalesced 512-byte
// load per wave, addressed without looking at the data, so the loads run ahead of the arithmetic; lanes whose lane element
// has run out sit the step out (execution mask), they cost no LDS traffic.
// A first blocked form kept the gather form's lane groups and CSR order (a group per row, column blocks of H in LDS): it
// was 30 % SLOWER than the gather form -- a row has ~5 non-zeros per block, so most of a group's 8..16 record slots per
// block were empty and the instruction count per non-zero went up, not down (profiles/r03/sparse_blocked.txt).
// ------------------------------------------------------------------------------------------------------
#ifdef NMFK_IS_F32
// OBJ: the non-zero terms of the objective (see sp_obj_kernel) instead of a half-step: rows of W as lane elements, Hobj the
// H to measure, the workgroup's partial in ossepart[1 + 4 tile] (the entries of the three other 256-row tiles are zeroed)
// SSE (round 4, the deferred check): a half-step (H orientation) that also leaves the objective's non-zero terms of the factors it
// reads -- its products p = <h_j, w_i> at the non-zeros are the ones sp_blk_obj_kernel recomputes -- in ossepart[1 + tile]
template <int NC, bool OBJ, bool SSE = false>
__device__ __forceinline__ void sp_blk_body(const NmfkSparseArgs &g, const NmfkRun &rd, const int tile, char *lds,
                                            const T *__restrict__ Hobj, double weight) {
  constexpr int KQ = 4 * NC;                   // signals a lane holds (kp rounded up to 4)
  constexpr int STR = NC == 1 ? 8 : (NC & 1) ? KQ + 8 : KQ + 4;  // = nmfk_spb_stride(NC): an odd number of 16-byte windows between rows
  constexpr int GPS = NMFK_SPB_LDS / (STR * 4) / NMFK_SPB_ROWS;  // = nmfk_spb_gps(NC)
  constexpr int P = NC > 4 ? 2 : 4;            // slot rows loaded ahead (four at 20..28 signals: no change, profiles/r03)
  constexpr int RSTEP = 1024 / KQ;             // rows staged per sweep of the workgroup (padded rows)
  const int kp = rd.kp, k = rd.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *__restrict__ Hcur = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it));
  T *__restrict__ Hnew = NMFK_PTR(T, g, NMFK_HOFF(rd, g.it + 1));
  T *__restrict__ Wt = NMFK_PTR(T, g, rd.oWt);
  const T *__restrict__ A = OBJ ? Wt : g.which == 0 ? Hcur : Wt;
  const T *__restrict__ B = OBJ ? Hobj : g.which == 0 ? Wt : Hnew;
  T *__restrict__ Anew = g.which == 0 ? Hnew : Wt;
  T *hb = (T *)lds;
  double sobj = 0;
  const int l = tile * NMFK_SPB_ROWS + tid;
  const bool valid = l < g.L;
  const bool vec = (kp & 3) == 0;  // (wave-uniform) rows are 16-byte multiples
  sp_vec4 a[NC], acc[NC];
  {
    const T *ar = A + (int64_t)(valid ? l : 0) * kp;
    if (vec) {
#pragma unroll
      for (int c = 0; c < NC; ++c) a[c] = *(const sp_vec4u *)(ar + 4 * c);
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[c][e] = 4 * c + e < kp ? ar[4 * c + e] : (T)0;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = (sp_vec4)((T)0);
  }
  const int ngb = g.ngb;
  const int slice = __builtin_amdgcn_readfirstlane(tile * 16 + wave);
  const bool has = slice < (g.L + 63) / 64;  // (the last workgroup's waves past the lane elements have no runs)
  const int32_t *__restrict__ ep = g.ellptr + (int64_t)(has ? slice : 0) * ngb;
  for (int gb0 = 0; gb0 < ngb; gb0 += GPS) {
    const int c0 = gb0 * NMFK_SPB_ROWS, rows = min(GPS * NMFK_SPB_ROWS, g.D - c0);
    __syncthreads();  // (everybody is done with the previous block)
    // (all loads of a thread first, then its LDS writes: one load -> wait -> write per trip left the latency of every
    //  trip exposed -- 3.6 us per granule where the L1 fill rate allows 1)
    if (vec) {
      const sp_vec4u *src = (const sp_vec4u *)(B + (int64_t)c0 * kp);
      // loads in flight per thread: 28..32 signals leave registers for two (four or eight in flight spill into the record
      // loop: 3.4 -> 4.4 -> 5.2 ms per H half-step of k = 17:32)
      constexpr int SB = NC >= 7 ? 2 : NC > 4 ? 4 : GPS * NC;
#pragma unroll 1
      for (int i0 = 0; i0 < GPS * NC; i0 += SB) {
        sp_vec4 tmp[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const int e = tid + (i0 + i) * 1024;
          if (i0 + i < GPS * NC && e < rows * NC) tmp[i] = src[e];
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const int e = tid + (i0 + i) * 1024, r = e / NC, c = e - r * NC;
          if (i0 + i < GPS * NC && e < rows * NC) *(sp_vec4 *)(hb + r * STR + 4 * c) = tmp[i];
        }
      }
    } else {  // pad the rows to KQ values with zeros
      const int c = tid % KQ, r0 = tid / KQ;
      if (tid < RSTEP * KQ) {
        constexpr int SS = NC >= 7 ? 4 : 8;
        for (int rb = r0; rb < rows; rb += SS * RSTEP) {
          T tmp[SS];
#pragma unroll
          for (int i = 0; i < SS; ++i) {
            const int r = min(rb + i * RSTEP, rows - 1);
            tmp[i] = B[(int64_t)(c0 + r) * kp + min(c, kp - 1)];
          }
#pragma unroll
          for (int i = 0; i < SS; ++i) tmp[i] = c < kp ? tmp[i] : (T)0;
#pragma unroll
          for (int i = 0; i < SS; ++i) {
            const int r = rb + i * RSTEP;
            if (r < rows) hb[r * STR + c] = tmp[i];
          }
        }
      }
    }
    __syncthreads();
    const int gb1 = min(gb0 + GPS, ngb);
    for (int gb = gb0; gb < gb1; ++gb) {
      const int base = ep[gb], len = has ? ep[gb + 1] - base : 0;  // (wave-uniform)
      if (len <= 0) continue;
      const int2 *__restrict__ e = g.ell + (int64_t)base * 64 + lane;
      int2 r[P];
#pragma unroll
      for (int j = 0; j < P; ++j) r[j] = e[j * 64];  // (past the run: the next run's records or the array's padding)
      for (int t = 0; t < len; t += P) {
        int2 cur[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
          cur[j] = r[j];
          r[j] = e[(t + P + j) * 64];
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
          if (t + j < len && cur[j].x >= 0) {  // (the first test is wave-uniform; the second masks lanes out)
            const T *hr = hb + (cur[j].x - c0) * STR;
            const T x = __builtin_bit_cast(float, cur[j].y);
            sp_vec4 h[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) h[c] = *(const sp_vec4 *)(hr + 4 * c);
            T2 s0 = T2{a[0][0], a[0][1]} * T2{h[0][0], h[0][1]}, s1 = T2{a[0][2], a[0][3]} * T2{h[0][2], h[0][3]};
#pragma unroll
            for (int c = 1; c < NC; ++c) {
              s0 = fma2(T2{a[c][0], a[c][1]}, T2{h[c][0], h[c][1]}, s0);
              s1 = fma2(T2{a[c][2], a[c][3]}, T2{h[c][2], h[c][3]}, s1);
            }
            const T pr = (s0.x + s0.y) + (s1.x + s1.y);
            if (OBJ || SSE) {
              const double xd = (double)x, p = (double)pr;
              sobj += (xd - p) * (xd - p) - p * p;
            }
            if (!OBJ) {
              const T q = div_t(x, pr);
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                const T2 lo = fma2(splat2(q), T2{h[c][0], h[c][1]}, T2{acc[c][0], acc[c][1]});
                const T2 hi = fma2(splat2(q), T2{h[c][2], h[c][3]}, T2{acc[c][2], acc[c][3]});
                acc[c] = sp_vec4{lo.x, lo.y, hi.x, hi.y};
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();  // the staged block is dead: its LDS serves the denominators, then the sums of the new values
  if (OBJ) {  // lanes of a wave in a butterfly (the same bits in every lane), then the 16 waves in order
    double *wsum = (double *)lds;
    sobj *= weight * weight;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sobj += __shfl_xor(sobj, o, 64);
    if (lane == 0) wsum[wave] = sobj;
    __syncthreads();
    if (tid < 4) {
      double t = 0;
      if (tid == 0)
        for (int wv = 0; wv < 16; ++wv) t += wsum[wv];
      const int slot = 4 * tile + tid;
      if (slot < (g.L + NMFK_TILE - 1) / NMFK_TILE) NMFK_PTR(double, g, rd.ossepart)[1 + slot] = t;
    }
    return;
  }
  if (SSE) {  // the workgroup's partial of the objective (lanes in a butterfly, the 16 waves in order), slot 1 + tile
    double *wsum = (double *)lds;
    sobj *= g.objw * g.objw;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sobj += __shfl_xor(sobj, o, 64);
    if (lane == 0) wsum[wave] = sobj;
    __syncthreads();
    double *part = NMFK_PTR(double, g, rd.ossepart);
    if (tid == 0) {
      double t = 0;
      for (int wv = 0; wv < 16; ++wv) t += wsum[wv];
      part[1 + tile] = t;
    }
    if (tile == 0)  // (the entries behind the lane tiles' own: check_a adds ntile_obj of them)
      for (int e = 1 + (g.L + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS + tid; e < 1 + g.ntile_obj; e += 1024) part[e] = 0.0;
    __syncthreads();
  }
  // denominators: the other factor's sum table, thread (j, c) adds the slots j, j + 32, ... of signal c, then the 32 partial
  // sums are added in order (a fixed order: reproducible)
  double *red = (double *)lds, *den = red + 1024;
  {
    const double *sumB = NMFK_PTR(const double, g, g.which == 0 ? rd.osumW : rd.osumH);
    const int PB = g.which == 0 ? rd.nsW : rd.nsH;
    const int c = tid & 31, j = tid >> 5;
    double sd = 0;
    if (c < kp)
      for (int q = j; q < PB; q += 32) sd += sumB[q * kp + c];
    red[tid] = sd;
    __syncthreads();
    if (tid < kp) {
      sd = 0;
      for (int jj = 0; jj < 32; ++jj) sd += red[jj * 32 + tid];
      den[tid] = sd;
    }
    __syncthreads();
  }
  // fused finish (Mult:67 / Mult:70 order)
  const T floorv = (g.clampw > g.it + 1 && g.which == 1 && (g.it + 1) % 10 == 0) ? (T)2.220446049250313e-16 : -(T)INFINITY;
  sp_vec4 v[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int sig = 4 * c + e;
      v[c][e] = (valid && sig < k) ? a[c][e] * acc[c][e] / (T)den[sig < kp ? sig : 0] : (T)0;
      if (valid && sig < k && v[c][e] < floorv) v[c][e] = floorv;  // (NmfkSparseArgs::clampw; a NaN stays)
    }
  if (valid) {
    T *dst = Anew + (int64_t)l * kp;
    if (vec) {
#pragma unroll
      for (int c = 0; c < NC; ++c) *(sp_vec4u *)(dst + 4 * c) = v[c];
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * c + e < kp) dst[4 * c + e] = v[c][e];
    }
  }
  // sums of the new values over the workgroup's lane elements -> slot `tile`: through LDS ([1024][KQ + 1] values behind
  // den), thread (j, c) adds the lane elements 32 j .. 32 j + 31 of signal c in fp64, then the 32 partial sums in order
  __syncthreads();  // (den has been read)
  T *vt = (T *)(lds + 1024 * sizeof(double));
  double *red2 = (double *)lds;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) vt[tid * (KQ + 1) + 4 * c + e] = v[c][e];
  __syncthreads();
  {
    const int c = tid & 31, j = tid >> 5;
    double sd = 0;
    if (c < kp)
      for (int r = 0; r < 32; ++r) sd += (double)vt[(32 * j + r) * (KQ + 1) + c];
    red2[tid] = sd;
    __syncthreads();
    if (tid < kp && tile < (g.which == 0 ? g.PH : g.PW)) {
      sd = 0;
      for (int jj = 0; jj < 32; ++jj) sd += red2[jj * 32 + tid];
      NMFK_PTR(double, g, g.which == 0 ? rd.osumH : rd.osumW)[(int64_t)tile * kp + tid] = sd;
    }
  }
}

// Which (tile, unit) a workgroup serves.  A workgroup streams its tile's share of the sliced ELL and the whole gathered
// factor of its unit; workgroups go round-robin over the 8 XCDs in linear-id order, so linear id -> (xcd, slot) and the
// slots of an XCD run through the tiles of one unit, then its next unit: a unit's gathered factor is fetched into ONE L2
// (0.5 MB of H stays there across the W half-step's tiles; the H half-step's tiles stream the 12.8 MB of W side by side).
// (Letting U units share a tile's records side by side instead -- U = 2, 4, 8 -- changed nothing: profiles/r03.)
__device__ __forceinline__ bool sp_blk_where(const NmfkSparseArgs &g, int cnt, int &tile, int &unit) {
  const int nt = (g.L + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  tile = slot % nt;
  unit = (slot / nt) * 8 + xcd;
  return unit < cnt;
}
// workgroups of a launch over cnt units
static int sp_blk_grid(const NmfkSparseArgs &a, int cnt) {
  return (a.L + NMFK_SPB_ROWS - 1) / NMFK_SPB_ROWS * ((cnt + 7) / 8 * 8);
}

__global__ __launch_bounds__(1024) void sp_blk_kernel(NmfkSparseArgs g, int u0, int cnt) {
  extern __shared__ char spb_lds[];
  int tile, ul;
  if (!sp_blk_where(g, cnt, tile, ul)) return;
  const int u = u0 + ul;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  switch ((rd.kp + 3) >> 2) {  // signals per lane / 4
    case 1: sp_blk_body<1, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 2: sp_blk_body<2, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 3: sp_blk_body<3, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 4: sp_blk_body<4, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 5: sp_blk_body<5, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 6: sp_blk_body<6, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 7: sp_blk_body<7, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 8: sp_blk_body<8, false>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    default: break;
  }
}
// the H half-step behind a check iteration (deferred check): the half-step and the objective's non-zero terms
__global__ __launch_bounds__(1024) void sp_blk_sse_kernel(NmfkSparseArgs g, int u0, int cnt) {
  extern __shared__ char spb_lds[];
  int tile, ul;
  if (!sp_blk_where(g, cnt, tile, ul)) return;
  const int u = u0 + ul;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  switch ((rd.kp + 3) >> 2) {
    case 1: sp_blk_body<1, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 2: sp_blk_body<2, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 3: sp_blk_body<3, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 4: sp_blk_body<4, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 5: sp_blk_body<5, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 6: sp_blk_body<6, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 7: sp_blk_body<7, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    case 8: sp_blk_body<8, false, true>(g, rd, tile, spb_lds, nullptr, 0.0); break;
    default: break;
  }
}
#ifdef NMFK_SPB_PROBE  // per-body register/spill counts (scripts/kernel_resources.py sp_blk_probe)
template <int NC>
__global__ __launch_bounds__(1024) void sp_blk_probe_kernel(NmfkSparseArgs g, int u0, int cnt) {
  extern __shared__ char spb_lds[];
  int tile, ul;
  if (!sp_blk_where(g, cnt, tile, ul)) return;
  sp_blk_body<NC, false>(g, g.runs[u0 + ul], tile, spb_lds, nullptr, 0.0);
}
template __global__ void sp_blk_probe_kernel<3>(NmfkSparseArgs, int, int);
template __global__ void sp_blk_probe_kernel<4>(NmfkSparseArgs, int, int);
template __global__ void sp_blk_probe_kernel<5>(NmfkSparseArgs, int, int);
template __global__ void sp_blk_probe_kernel<6>(NmfkSparseArgs, int, int);
template __global__ void sp_blk_probe_kernel<7>(NmfkSparseArgs, int, int);
template __global__ void sp_blk_probe_kernel<8>(NmfkSparseArgs, int, int);
#endif
// the objective's non-zero terms of the units of ranks up to 32 (the others leave at once: sp_obj_kernel serves them)
__global__ __launch_bounds__(1024) void sp_blk_obj_kernel(NmfkSparseArgs g, int hsel, int total_iters, double weight, int u0,
                                                          int cnt) {
  extern __shared__ char spb_lds[];
  int tile, ul;
  if (!sp_blk_where(g, cnt, tile, ul)) return;
  const int u = u0 + ul;
  const NmfkState st = g.state[u];
  if (!g.force && !st.active) return;
  const NmfkRun rd = g.runs[u];
  if (!nmfk_sp_blk_rank(rd.kp)) return;
  const int sel = hsel >= 0 ? hsel : ((st.active ? total_iters : st.iters) & 1);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, sel));
  switch ((rd.kp + 3) >> 2) {
    case 1: sp_blk_body<1, true>(g, rd, tile, spb_lds, H, weight); break;
    case 2: sp_blk_body<2, true>(g, rd, tile, spb_lds, H, weight); break;
    case 3: sp_blk_body<3, true>(g, rd, tile, spb_lds, H, weight); break;
    case 4: sp_blk_body<4, true>(g, rd, tile, spb_lds, H, weight); break;
    case 5: sp_blk_body<5, true>(g, rd, tile, spb_lds, H, weight); break;
    case 6: sp_blk_body<6, true>(g, rd, tile, spb_lds, H, weight); break;
    case 7: sp_blk_body<7, true>(g, rd, tile, spb_lds, H, weight); break;
    case 8: sp_blk_body<8, true>(g, rd, tile, spb_lds, H, weight); break;
    default: break;
  }
}
#endif

// objective on sparse X:  sum_all (x - p)^2 = sum_nz [(x - p)^2 - p^2] + sum_all p^2,  sum_all p^2 = <W'W, HH'>.
// part 1: the non-zero terms, the same walk over the rows (CSR) as the W half-step, fp64 accumulation, one partial per
// workgroup (= 256 rows) in ossepart[1 + tile]
template <int LPR>
__device__ __forceinline__ void sp_obj_body(const NmfkSparseArgs &g, const NmfkRun &rd, const T *__restrict__ H, double weight,
                                            double *sh) {
  constexpr int GPW = NMFK_TILE / LPR;
  const int tid = threadIdx.x, grp = tid / LPR;
  const T *__restrict__ Wt = NMFK_PTR(const T, g, rd.oWt);
  SpWalk<LPR> w;
  w.init(rd.kp, tid);
  double s = 0;
#pragma unroll 1
  for (int pass = 0; pass < LPR; ++pass) {
    const int i = blockIdx.x * NMFK_TILE + pass * GPW + grp;
    const bool valid = i < g.L;
    const int ic = valid ? i : 0;
    const sp_vec4 a = w.lane_row(Wt, ic);
    const int p0 = valid ? g.ptr[ic] : 0, p1 = valid ? g.ptr[ic + 1] : 0;
    w.walk(g, H, a, p0, p1, [&](int, bool live, T x, T pr, const sp_vec4 &) __attribute__((always_inline)) {
      const double xd = (double)x, p = (double)pr;
      s += live ? (xd - p) * (xd - p) - p * p : 0.0;
    });
  }
  if (w.sub != 0) s = 0;  // every lane of a group holds the group's terms
  s = block_sum(s * weight * weight, sh);
  if (threadIdx.x == 0) NMFK_PTR(double, g, rd.ossepart)[1 + blockIdx.x] = s;
}
__global__ __launch_bounds__(NMFK_TILE) void sp_obj_kernel(NmfkSparseArgs g, int hsel, int total_iters, double weight, int u0) {
  __shared__ double sh[8];
  const int u = u0 + blockIdx.y;
  const NmfkState st = g.state[u];
  if (!g.force && !st.active) return;
  const NmfkRun rd = g.runs[u];
#ifdef NMFK_IS_F32
  if (g.ell && nmfk_sp_blk_rank(rd.kp)) return;  // (sp_blk_obj_kernel serves it)
#endif
  const int sel = hsel >= 0 ? hsel : ((st.active ? total_iters : st.iters) & 1);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, sel));
  if (rd.kp <= 4)
    sp_obj_body<1>(g, rd, H, weight, sh);
  else if (rd.kp <= 8)
    sp_obj_body<2>(g, rd, H, weight, sh);
  else if (rd.kp <= 16)
    sp_obj_body<4>(g, rd, H, weight, sh);
  else if (rd.kp <= 32)
    sp_obj_body<8>(g, rd, H, weight, sh);
  else
    sp_obj_body<16>(g, rd, H, weight, sh);
}

// part 2: <W'W, HH'> -> ssepart[0].  Stage 1 (grid: row chunks of W then of H, units): the partial Gram matrix of
// NMFK_GRAM_ROWS rows of a factor on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64: A = 4 rows x 16 signals transposed,
// B = the same rows x 16 signals, one 16 x 16 block of F'F per wave task), written as it leaves the accumulators:
// both factors use the same layout and the inner product of stage 2 is elementwise, so the layout never matters.
typedef double sp_f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(NMFK_TILE) void sp_gram_part_kernel(NmfkSparseArgs g, int n, int m, int hsel, int total_iters,
                                                                int u0) {
  const int u = u0 + blockIdx.y;
  const NmfkState st = g.state[u];
  if (!g.force && !st.active) return;
  const NmfkRun rd = g.runs[u];
  const int sel = hsel >= 0 ? hsel : ((st.active ? total_iters : st.iters) & 1);
  const int kp = rd.kp, nb = (kp + 15) / 16, nblk = nb * nb, rp = nblk >= 4 ? 1 : 4;
  const int cw = (n + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS;
  const bool isW = (int)blockIdx.x < cw;
  const T *__restrict__ F = isW ? NMFK_PTR(const T, g, rd.oWt) : NMFK_PTR(const T, g, NMFK_HOFF(rd, sel));
  const int len = isW ? n : m, cidx = isW ? blockIdx.x : blockIdx.x - cw;
  const int r0 = cidx * NMFK_GRAM_ROWS, r1 = min(len, r0 + NMFK_GRAM_ROWS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, lq = lane >> 4;
  double *out = NMFK_PTR(double, g, rd.ogram);
  for (int task = wave; task < nblk * rp; task += NMFK_TILE / 64) {
    const int blk = task % nblk, part = task / nblk, bi = blk / nb, bj = blk - bi * nb;
    const int q = ((((r1 - r0) + rp - 1) / rp) + 3) & ~3;
    const int s0 = min(r1, r0 + part * q), s1 = min(r1, s0 + q);
    const int ca = 16 * bi + l16, cb = 16 * bj + l16;
    const bool va = ca < kp, vb = cb < kp;
    const T *__restrict__ Fa = F + (va ? ca : 0), *__restrict__ Fb = F + (vb ? cb : 0);
    sp_f64x4 acc = {0, 0, 0, 0};
    constexpr int UN = 8;
    for (int r = s0; r < s1; r += 4 * UN) {
      T av[UN], bv[UN];
#pragma unroll
      for (int i = 0; i < UN; ++i) {
        const int row = min(r + 4 * i + lq, len - 1);
        av[i] = Fa[(int64_t)row * kp];
        bv[i] = Fb[(int64_t)row * kp];
      }
#pragma unroll
      for (int i = 0; i < UN; ++i) {
        const bool ok = r + 4 * i + lq < s1;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && va) ? (double)av[i] : 0.0, (ok && vb) ? (double)bv[i] : 0.0, acc, 0, 0, 0);
      }
    }
    const int pidx = (isW ? 0 : cw * rp) + cidx * rp + part;
    *(sp_f64x4 *)(out + ((int64_t)pidx * nblk + blk) * 256 + lane * 4) = acc;
  }
}
// stage 2 (one workgroup per unit): the partial matrices of each factor added in order, then their inner product
__global__ __launch_bounds__(NMFK_TILE) void sp_gram_dot_kernel(NmfkSparseArgs g, int n, int m, double weight, int u0) {
  __shared__ double sh[8];
  const int u = u0 + blockIdx.x;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  const int kp = rd.kp, nb = (kp + 15) / 16, nblk = nb * nb, rp = nblk >= 4 ? 1 : 4;
  const int npw = ((n + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS) * rp, nph = ((m + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS) * rp;
  const double *part = NMFK_PTR(const double, g, rd.ogram);
  const int64_t msz = (int64_t)nblk * 256;
  double total = 0;
  for (int e = threadIdx.x; e < msz; e += NMFK_TILE) {
    double gw = 0, gh = 0;
    for (int q = 0; q < npw; ++q) gw += part[q * msz + e];
    for (int q = 0; q < nph; ++q) gh += part[(npw + q) * msz + e];
    total += gw * gh;
  }
  total = block_sum(total * weight * weight, sh);
  if (threadIdx.x == 0) NMFK_PTR(double, g, rd.ossepart)[0] = total;
}

// ------------------------------------------------------------------------------------------------------
// half-step finish: A_new = A .* (sum of partial numerators) ./ sumB ;  sumA_new.  Grid (slots of A's table, units).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void reduce_kernel(NmfkStepArgs g, int u0) {
  __shared__ double sh[NMFK_TILE];
  __shared__ double den[NMFK_MAX_K];
  const int u = u0 + blockIdx.y, b = blockIdx.x;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  const T *Aold;
  T *Anew;
  const double *sumB;
  double *sumA;
  int PA, PB;
  if (g.which == 0) {
    Aold = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it));
    Anew = NMFK_PTR(T, g, NMFK_HOFF(rd, g.it + 1));
    sumB = NMFK_PTR(const double, g, rd.osumW);
    sumA = NMFK_PTR(double, g, rd.osumH);
    PB = rd.nsW;  // (the slots behind the unit's own are zero)
    PA = rd.nsH;
  } else {
    Aold = NMFK_PTR(const T, g, rd.oWt);
    Anew = NMFK_PTR(T, g, rd.oWt);
    sumB = NMFK_PTR(const double, g, rd.osumH);
    sumA = NMFK_PTR(double, g, rd.osumW);
    PB = rd.nsH;
    PA = rd.nsW;
  }
  const int k = rd.k, kp = rd.kp;
  if (b >= PA) {
    zero_slot(sumA + (int64_t)b * kp, kp);
    return;
  }
  if (threadIdx.x < kp) den[threadIdx.x] = nmfk_slot_sum<16>(sumB, kp, PB, threadIdx.x);  // (256 slots of W at 65536 rows)
  __syncthreads();
  int l0, l1;
  slot_range(g.L, PA, b, l0, l1);
  const T *part = NMFK_PTR(const T, g, rd.opart);
  const int64_t LK = (int64_t)g.L * kp;
  const T floorv = (g.clampw > g.it + 1 && g.which == 1 && (g.it + 1) % 10 == 0) ? (T)2.220446049250313e-16 : -(T)INFINITY;
  for (int64_t e = (int64_t)l0 * kp + threadIdx.x; e < (int64_t)l1 * kp; e += NMFK_TILE) {
    const int c = (int)(e % kp);
    const T aold = Aold[e];
    T num = (T)0;
    for (int s0 = 0; s0 < g.S; s0 += 8) {  // eight partials in flight, added in the order of the splits
      T pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s0 + j < g.S ? part[(int64_t)(s0 + j) * LK + e] : (T)0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s0 + j < g.S) num += pv[j];
    }
    T v = aold * num / (T)den[c];  // same operation order as Mult:67,70
    v = v < floorv ? floorv : v;      // (NmfkStepArgs::clampw: W of a check iteration; a NaN stays)
    if (c >= k) v = (T)0;
    Anew[e] = v;
  }
  __syncthreads();
  range_signal_sums(Anew, kp, k, l0, l1, sumA + (int64_t)b * kp, sh);
}

// ------------------------------------------------------------------------------------------------------
// objective partials: sum((((X - W*H) .* weight)[.!inan]).^2)  (Mult:74) / normnan (Exec:791-792)
// lanes = rows of X (column-major copy), loop over columns; fp64 accumulation.
// ------------------------------------------------------------------------------------------------------
template <int KP>
__device__ __forceinline__ void sse_body(const NmfkSseArgs &g, const NmfkRun &rd, const T *__restrict__ H, double *sh) {
  const int i = blockIdx.x * NMFK_TILE + threadIdx.x;
  const bool valid = i < g.n;
  const int ic = valid ? i : 0;
  const T *__restrict__ Wt = NMFK_PTR(const T, g, rd.oWt);
  T a[KP];
#pragma unroll
  for (int c = 0; c < KP; ++c) a[c] = Wt[c + (int64_t)ic * KP];
  const float *__restrict__ xp = g.Xc + ic;
  const T wgt = (T)g.weight;
  double ssum = 0.0;
#pragma unroll 4
  for (int j = 0; j < g.m; ++j) {
    const T *__restrict__ b = H + (int64_t)j * KP;
    T p = (T)0;
#pragma unroll
    for (int c = 0; c < KP; ++c) p = fma_t(a[c], b[c], p);
    const float xf = xp[(int64_t)j * g.n];
    const T wij = g.Wgt ? wgt * (T)g.Wgt[ic + (int64_t)j * g.n] : wgt;
    const T e = ((T)xf - p) * wij;
    const double e2 = (double)e * (double)e;
    bool use = valid && (xf == xf);
    if (g.force) use = use && (e == e);  // normnan skips NaN residuals too (Help:226-228)
    ssum += use ? e2 : 0.0;
  }
  ssum = block_sum(ssum, sh);
  if (threadIdx.x == 0) NMFK_PTR(double, g, rd.ossepart)[blockIdx.x] = ssum;
}

// fp32 compute, no weight array, not the final normnan pass: two rows per thread as packed pairs (v_pk_fma_f32 with the
// scalar H entries), residuals squared in fp32 and added in fp32 over FOUR columns before they enter the fp64 sum.
// (A partial of 8 squares of ~0.1 carries ~1e-7 relative rounding noise, unbiased; over the 4.2e6 entries of the
// BASELINE shape that is ~3e-10 of the objective, 1e-4 absolute on 3.5e5 -- the stop rule's tolOF is 1e-3.  The
// per-element fp64 version cost as much as 4 half-steps per check at k <= 8: 20 % of the packed-VALU ranks' time.)
// One partial per workgroup = 512 rows: slot 2b, slot 2b + 1 is zeroed.
template <int KP>
__device__ __forceinline__ void sse_body_pk(const NmfkSseArgs &g, const NmfkRun &rd, const float *__restrict__ H, double *sh) {
  const int tid = threadIdx.x, i0 = blockIdx.x * 2 * NMFK_TILE + tid, i1 = i0 + NMFK_TILE;
  const bool v0 = i0 < g.n, v1 = i1 < g.n;
  const int c0 = v0 ? i0 : 0, c1 = v1 ? i1 : 0;
  const float *__restrict__ Wt = NMFK_PTR(const float, g, rd.oWt);
  f32x2 a2[KP];
#pragma unroll
  for (int c = 0; c < KP; ++c) a2[c] = (f32x2){Wt[c + (int64_t)c0 * KP], Wt[c + (int64_t)c1 * KP]};
  const float *__restrict__ xp0 = g.Xc + c0, *__restrict__ xp1 = g.Xc + c1;
  const float w = (float)g.weight;
  const f32x2 w2 = {v0 ? w : 0.f, v1 ? w : 0.f};  // rows past the end contribute nothing
  double ssum = 0.0;
  auto column = [&](int j) __attribute__((always_inline)) -> float {
    const float *__restrict__ b = H + (int64_t)j * KP;
    f32x2 p2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KP; ++c) p2 = __builtin_elementwise_fma((f32x2)(b[c]), a2[c], p2);  // (broadcast first: see fma2)
    const f32x2 x2 = {xp0[(int64_t)j * g.n], xp1[(int64_t)j * g.n]};
    f32x2 e2 = (x2 - p2) * w2;
    e2.x = x2.x == x2.x ? e2.x : 0.f;  // missing entries (NaN) do not count
    e2.y = x2.y == x2.y ? e2.y : 0.f;
    const f32x2 q = e2 * e2;
    return q.x + q.y;
  };
  int j = 0;
  for (; j + 4 <= g.m; j += 4) {
    float s4 = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) s4 += column(j + jj);
    ssum += (double)s4;
  }
  for (; j < g.m; ++j) ssum += (double)column(j);
  ssum = block_sum(ssum, sh);
  if (tid == 0) {
    double *part = NMFK_PTR(double, g, rd.ossepart);
    part[2 * blockIdx.x] = ssum;
    if (2 * (int)blockIdx.x + 1 < (g.n + NMFK_TILE - 1) / NMFK_TILE) part[2 * blockIdx.x + 1] = 0.0;
  }
}

#define NMFK_SSE_CASE(KP) sse_body<KP>(g, rd, H, sh)
__global__ __launch_bounds__(NMFK_TILE) void sse_kernel(NmfkSseArgs g, int u0) {
  __shared__ double sh[8];
  const int u = u0 + blockIdx.y;
  const NmfkState st = g.state[u];
  if (!g.force && !st.active) return;
  const NmfkRun rd = g.runs[u];
  const int sel = g.hsel >= 0 ? g.hsel : ((st.active ? g.total_iters : st.iters) & 1);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, sel));
  NMFK_DISPATCH_KP(rd.kp, NMFK_SSE_CASE)
}
#define NMFK_SSEPK_CASE(KP) sse_body_pk<KP>(g, rd, (const float *)H, sh)
__global__ __launch_bounds__(NMFK_TILE) void sse_pk_kernel(NmfkSseArgs g, int u0) {
  __shared__ double sh[8];
  const int u = u0 + blockIdx.y;
  const NmfkState st = g.state[u];
  if (!st.active) return;
  const NmfkRun rd = g.runs[u];
  const int sel = g.hsel >= 0 ? g.hsel : ((st.active ? g.total_iters : st.iters) & 1);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, sel));
  if (sizeof(T) == 4) {
    NMFK_DISPATCH_KP(rd.kp, NMFK_SSEPK_CASE)
  }
}

// out[u] = sum of the objective partials of unit u (fixed order)
__global__ void sum_parts_kernel(char *arena, const NmfkRun *runs, int nunits, int ntile, double *out) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nunits) return;
  const double *part = (const double *)(arena + runs[u].ossepart);
  double s = 0;
  for (int t = 0; t < ntile; ++t) s += part[t];
  out[u] = s;
}

// ------------------------------------------------------------------------------------------------------
// check block, every 10th iteration (Mult:73-117), three launches in stream order:
//   check_a  objective -> tol test -> bad-iteration bookkeeping           (one thread per unit)
//   clamp    H = max.(H, eps()), W = max.(W, eps()) and the sum tables    (grid (slots, units))
//   check_b  co-clustering consistency -> loop guard (Mult:64)            (one workgroup per unit)
// ------------------------------------------------------------------------------------------------------
__global__ void check_a_kernel(NmfkCheckArgs g, int u0, int cnt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  const int u = u0 + q;
  NmfkState *st = g.state + u;
  if (!st->active) return;
  const NmfkRun rd = g.runs[u];
  double obj = 0;
  for (int t = 0; t < g.ntile_n; ++t) obj += NMFK_PTR(const double, g, rd.ossepart)[t];
  st->last_obj = obj;
  if (g.trace && (g.it + 1) / 10 - 1 < g.trace_stride) g.trace[(int64_t)rd.uid * g.trace_stride + (g.it + 1) / 10 - 1] = obj;
  if (obj < g.tol) {  // Mult:75-78: leaves the loop before the clamp
    st->active = 0;
    st->reason = NMFK_STOP_TOL;
    st->iters = g.it + 1;
  } else {  // Mult:79-98
    double best = st->best;
    int bad = st->baditers, re = st->reattempts;
    if (obj < best) {
      if ((best - obj) < g.tolOF)
        bad += 1;
      else
        bad = 0;
      best = obj;
    } else {
      bad += 1;
    }
    if (bad >= g.maxbaditers) {
      re += 1;
      bad = 0;
    }
    st->best = best;
    st->baditers = bad;
    st->reattempts = re;
  }
}

// Mult:99-100; eps() is Float64 eps whatever T is; NaN stays NaN
__global__ __launch_bounds__(NMFK_TILE) void clamp_kernel(NmfkCheckArgs g, int u0) {
  __shared__ double sh[NMFK_TILE];
  const int u = u0 + blockIdx.y, b = blockIdx.x;
  if (!g.state[u].active) return;
  // nothing below eps() was written since the last clamp (the fused finishes of the check iteration watch their values): the
  // factors and their sum tables are already what this pass would leave
  if (g.track_low && !g.state[u].lowflag) return;
  const NmfkRun rd = g.runs[u];
  const int tid = threadIdx.x, k = rd.k, kp = rd.kp;
  const T eps = (T)2.220446049250313e-16;
  for (int f = g.w_clamped ? 1 : 0; f < 2; ++f) {
    const int P = f == 0 ? rd.nsW : rd.nsH, PT = f == 0 ? g.PW : g.PH, len = f == 0 ? g.n : g.m;
    T *F = f == 0 ? NMFK_PTR(T, g, rd.oWt) : NMFK_PTR(T, g, NMFK_HOFF(rd, g.it + 1));
    double *tab = NMFK_PTR(double, g, f == 0 ? rd.osumW : rd.osumH);
    if (b >= P) {
      if (b < PT) zero_slot(tab + (int64_t)b * kp, kp);
      continue;
    }
    int l0, l1;
    slot_range(len, P, b, l0, l1);
    range_clamp_sums(F, kp, k, l0, l1, eps, tab + (int64_t)b * kp, sh);  // (one pass over the slot's rows; round 4)
  }
}

__global__ __launch_bounds__(NMFK_TILE) void check_b_kernel(NmfkCheckArgs g, int u0) {
  __shared__ int sh_diff;
  __shared__ int sh_first[NMFK_MAX_K];
  const int u = u0 + blockIdx.x;
  NmfkState *st = g.state + u;
  if (!st->active) return;
  const NmfkRun rd = g.runs[u];
  const int tid = threadIdx.x, k = rd.k, kp = rd.kp, m = g.m;
  if (tid == 0) sh_diff = 0;
  if (tid < NMFK_MAX_K) sh_first[tid] = 0x7fffffff;
  __syncthreads();
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it + 1));

  // index[q] = argmin(H[:,q]) (first minimum; a NaN wins, as in Julia); cons[i,j] = index[i]==index[j];
  // consdiff == 0  <=>  the partition of the columns is unchanged.  Canonical form of a partition: every
  // column labelled by the first column of its class.
  int32_t *idx = NMFK_PTR(int32_t, g, rd.ocanon) + m;  // scratch behind the partition (the partial-numerator buffer may hold the H
                                                       // half-step's partials for the W half-step to sum: NmfkStepArgs::fuse_red)
  for (int q = tid; q < m; q += NMFK_TILE) {
    int am = 0;
    T best = H[(int64_t)q * kp];
    bool bnan = best != best;
    for (int c = 1; c < k; ++c) {
      const T v = H[c + (int64_t)q * kp];
      if (!bnan && (v != v || v < best)) {
        best = v;
        am = c;
        bnan = v != v;
      }
    }
    idx[q] = am;
    atomicMin(&sh_first[am], q);
  }
  __syncthreads();
  const int have_old = st->have_old;
  int32_t *canon = NMFK_PTR(int32_t, g, rd.ocanon);
  int diff = 0;
  for (int q = tid; q < m; q += NMFK_TILE) {
    const int cn = sh_first[idx[q]];
    if (!have_old || canon[q] != cn) diff = 1;
    canon[q] = cn;
  }
  if (diff) atomicOr(&sh_diff, 1);
  __syncthreads();
  if (tid == 0) {
    int inc = (have_old && !sh_diff) ? st->inc + 1 : 0;  // Mult:106-111 (first check always differs)
    st->inc = inc;
    if (g.track_low) st->lowflag = 0;  // (the clamp has run, or had nothing to do; the next check iteration's finishes set it again)
    st->have_old = 1;
    const int iters = g.it + 1;
    if (inc > g.stopconv) {  // Mult:112-115
      st->active = 0;
      st->reason = NMFK_STOP_CONSISTENCY;
      st->iters = iters;
    } else if (st->reattempts >= g.maxreattempts || st->baditers >= g.maxbaditers) {  // Mult:64
      st->active = 0;
      st->reason = NMFK_STOP_STAGNATION;
      st->iters = iters;
    } else if (iters >= g.maxiter) {
      st->active = 0;
      st->reason = NMFK_STOP_MAXITER;
      st->iters = iters;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// finish (Exec:790-805): objvalue = normnan(X - W*H); total = sum(H;dims=2); W .*= total'; H ./= total;
// results stored as T = Float32 (Exec:529-531).  Grid (slices of the rows of W / columns of H, units).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void finish_kernel(NmfkFinishArgs g) {
  __shared__ double sh[NMFK_TILE];
  __shared__ double rs[NMFK_MAX_K];
  const int u = blockIdx.y, b = blockIdx.x, NB = gridDim.x;
  NmfkState *st = g.state + u;
  const NmfkRun rd = g.runs[u];
  const int tid = threadIdx.x, k = rd.k, kp = rd.kp, n = g.n, m = g.m;
  int iters = st->iters, reason = st->reason;
  if (st->active) {  // the host loop ran out of iterations (maxiter % 10 != 0 or no check fired)
    iters = g.total_iters;
    reason = NMFK_STOP_MAXITER;
  }
  const T *Wt = NMFK_PTR(const T, g, rd.oWt);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, iters));
  range_signal_sums(H, kp, k, 0, m, rs, sh);
  float *Wo = g.Wout[rd.kidx] + (int64_t)rd.ridx * n * k;
  float *Ho = g.Hout[rd.kidx] + (int64_t)rd.ridx * m * k;
  int i0, i1, j0, j1;
  slot_range(n, NB, b, i0, i1);
  slot_range(m, NB, b, j0, j1);
  const int ni = i1 - i0;
  for (int64_t e = tid; e < (int64_t)ni * k; e += NMFK_TILE) {
    const int c = (int)(e / ni);
    const int64_t i = i0 + (e - (int64_t)c * ni);
    T v = Wt[c + i * kp];
    if (g.normalize) v = v * (T)rs[c];
    Wo[i + (int64_t)c * n] = (float)v;
  }
  for (int64_t e = (int64_t)j0 * k + tid; e < (int64_t)j1 * k; e += NMFK_TILE) {
    const int c = (int)(e % k);
    const int64_t j = e / k;
    T v = H[c + j * kp];
    if (g.normalize) v = v / (T)rs[c];
    Ho[e] = (float)v;
  }
  if (b == 0 && tid == 0) {
    double obj = 0;
    for (int t = 0; t < g.ntile_n; ++t) obj += NMFK_PTR(const double, g, rd.ossepart)[t];
    g.frob[rd.kidx][rd.ridx] = (float)sqrt(obj > 0 ? obj : 0.0);  // (the sparse form can round to a tiny negative)
    g.iters[rd.kidx][rd.ridx] = iters;
    g.reason[rd.kidx][rd.ridx] = reason;
  }
}

// after finish: the units' final state (separate launch: finish workgroups of one unit read it concurrently)
__global__ void finish_state_kernel(NmfkFinishArgs g) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= g.nunits) return;
  NmfkState *st = g.state + u;
  if (st->active) {
    st->iters = g.total_iters;
    st->reason = NMFK_STOP_MAXITER;
    st->active = 0;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// launchers
//
// Grid order of the half-step kernels.  Workgroups are dealt to the 8 XCDs round-robin in linear order.  Default
// (tile fastest): XCD j runs the lane tiles {j, j+8, ...} of EVERY unit of the launch, so its L2 (4 MB) only ever
// sees 1/8 of X (2.1 MB of the 16.8 MB at the bench shape) and serves it to all the restarts: PMC, k = 16, 32
// restarts per launch: TCC hit rate 91 %, 54 MB fetched from the fabric per launch.  NMFK_UNIT_FAST = 1 makes the
// unit the fast dimension (unit u on XCD u mod 8: the factors of a unit stay in one L2, but every XCD streams all of
// X): TCC hit rate 82 %, 79 MB per launch, same run time.  grid.y is limited to 65535.
// ------------------------------------------------------------------------------------------------------
static inline int nmfk_unit_fast(int tiles) { return (NMFK_UNIT_FAST != 0) && tiles <= 65535; }
#define NMFK_GRID(uf, tiles, cnt) ((uf) ? dim3((cnt), (tiles)) : dim3((tiles), (cnt)))
void NMFK_NAME(nmfk_launch_init)(const NmfkInitArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(init_kernel, dim3(std::max(a.PW, a.PH), a.nunits), dim3(NMFK_TILE), 0, s, a);
}

// `dargs`: device copy of `a` (constant over the sweep except `it`, which is passed by value).
// Launches the half-step of the `cnt` units [u0, u0 + cnt), which all have padded rank kp.
template <int KP>
static void launch_step_kp(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int u0, int cnt, hipStream_t s) {
  constexpr int LB = NMFK_LB_OF(KP);
  const int ws = a.wsplit;
  const int lpw = ws > 1 ? 64 : NMFK_TILE;
  const int ntile = (a.L + lpw * LB - 1) / (lpw * LB);
  const int uf = nmfk_unit_fast(ntile * a.S);
  const dim3 grid = NMFK_GRID(uf, ntile * a.S, cnt), blk(ws > 1 ? 64 * ws : NMFK_TILE);
  size_t scratch = ws > 1 ? (size_t)(ws - 1) * LB * KP * 64 * sizeof(T) : 0;
  const size_t ldsb = sizeof(double) * 9 * NMFK_MAX_K + scratch;
  if (a.has_nan)
    hipLaunchKernelGGL((step_kernel<true, KP>), grid, blk, ldsb, s, a.arena, a.X, a.runs, a.state, dargs, a.it, u0, uf);
  else
    hipLaunchKernelGGL((step_kernel<false, KP>), grid, blk, ldsb, s, a.arena, a.X, a.runs, a.state, dargs, a.it, u0, uf);
}

#define NMFK_LAUNCH_CASE(KP) launch_step_kp<KP>(a, dargs, u0, cnt, s)
void NMFK_NAME(nmfk_launch_step)(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt,
                                 hipStream_t s) {
  NMFK_DISPATCH_KP(kp, NMFK_LAUNCH_CASE)
}

#if !defined(NMFK_IS_F32) || NMFK_WITH_MERGED_F32
void NMFK_NAME(nmfk_launch_step_multi)(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int u0, int cnt, hipStream_t s) {
  constexpr int LB = NMFK_MULTI_LB;
  const int ws = a.wsplit;
  const int lpw = ws > 1 ? 64 : NMFK_TILE;
  const int ntile = (a.L + lpw * LB - 1) / (lpw * LB);
  const int uf = nmfk_unit_fast(ntile * a.S);
  const dim3 grid = NMFK_GRID(uf, ntile * a.S, cnt), blk(ws > 1 ? 64 * ws : NMFK_TILE);
  size_t scratch = ws > 1 ? (size_t)(ws - 1) * LB * 16 * 64 * sizeof(T) : 0;
  const size_t ldsb = sizeof(double) * 9 * NMFK_MAX_K + scratch;
  if (a.has_nan)
    hipLaunchKernelGGL((step_kernel_multi<true>), grid, blk, ldsb, s, a.arena, a.X, a.runs, a.state, dargs, a.it, u0, uf);
  else
    hipLaunchKernelGGL((step_kernel_multi<false>), grid, blk, ldsb, s, a.arena, a.X, a.runs, a.state, dargs, a.it, u0, uf);
}
#endif

#ifdef NMFK_IS_F32
// all-MFMA half-step for 16 < kp <= 64 (fp32, no missing data, D >= 16)
template <int KQ, int NB, int NT>
static void launch_mfma_wide(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int u0, int cnt, hipStream_t s) {
  const int ws = a.wsplit, nwaves = ws > 1 ? ws : 4;
  const int lpw = 16 * NT * (ws > 1 ? 1 : nwaves);
  const int ntile = (a.L + lpw - 1) / lpw;
  const int uf = nmfk_unit_fast(ntile * a.S);
  const dim3 grid = NMFK_GRID(uf, ntile * a.S, cnt), blk(64 * nwaves);
  const size_t stage = (size_t)nwaves * 16 * (4 * KQ + 4) * sizeof(float);
  const size_t cross = ws > 1 ? (size_t)(ws - 1) * NT * NB * 4 * 64 * sizeof(float) : 0;
  const size_t ldsb = sizeof(double) * 9 * NMFK_MAX_K + std::max(stage, cross);
  hipLaunchKernelGGL((mfma_wide_kernel<KQ, NB, NT>), grid, blk, ldsb, s, a.arena, a.X, a.runs, a.state, dargs, a.it, u0, uf);
}

// objective of ranks above 16 on the matrix pipe (fp32, dense, no missing data, scalar weight)
void nmfk_launch_sse_mfma_wide_f32(const NmfkSseArgs &a, int kp, int u0, int cnt, hipStream_t s) {
  const int ntile = (a.n + NMFK_TILE - 1) / NMFK_TILE;
  const dim3 grid(ntile, cnt), blk(NMFK_TILE);
  const size_t ldsb = sizeof(double) * 8 + sizeof(float) * 4 * 16 * (size_t)(kp + 4);
#define NMFK_SSE_MFMA(KQ) hipLaunchKernelGGL((mfma_sse_kernel<KQ>), grid, blk, ldsb, s, a, u0)
  switch (kp) {
    case 20: NMFK_SSE_MFMA(5); break;   case 24: NMFK_SSE_MFMA(6); break;   case 28: NMFK_SSE_MFMA(7); break;
    case 32: NMFK_SSE_MFMA(8); break;   case 40: NMFK_SSE_MFMA(10); break;  case 48: NMFK_SSE_MFMA(12); break;
    case 56: NMFK_SSE_MFMA(14); break;  case 64: NMFK_SSE_MFMA(16); break;
    default: break;
  }
#undef NMFK_SSE_MFMA
}

int nmfk_mfma_wide_lane_tile(int wsplit) { return 16 * NMFK_WIDE_NT * (wsplit > 1 ? 1 : 4); }

void nmfk_launch_step_mfma_wide_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt,
                                    hipStream_t s) {
  switch (kp) {
    case 20: launch_mfma_wide<5, 2, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 24: launch_mfma_wide<6, 2, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 28: launch_mfma_wide<7, 2, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 32: launch_mfma_wide<8, 2, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 40: launch_mfma_wide<10, 3, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 48: launch_mfma_wide<12, 3, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 56: launch_mfma_wide<14, 4, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    case 64: launch_mfma_wide<16, 4, NMFK_WIDE_NT>(a, dargs, u0, cnt, s); break;
    default: break;
  }
}
#endif

void NMFK_NAME(nmfk_launch_sp_step)(const void *argsv, int kp, int u0, int cnt, hipStream_t s) {
  const NmfkSparseArgs &a = *(const NmfkSparseArgs *)argsv;
#ifdef NMFK_IS_F32
  if (a.ell && nmfk_sp_blk_rank(kp)) {  // blocked form: a lane element per thread, the gathered factor through LDS
    static std::atomic<uint64_t> lds_ok{0}, lds_oks{0};
    if (a.objw > 0 && a.which == 0) {
      nmfk_allow_dynamic_lds((const void *)sp_blk_sse_kernel, lds_oks, NMFK_SPB_LDS);
      hipLaunchKernelGGL(sp_blk_sse_kernel, dim3(sp_blk_grid(a, cnt)), dim3(1024), NMFK_SPB_LDS, s, a, u0, cnt);
    } else {
      nmfk_allow_dynamic_lds((const void *)sp_blk_kernel, lds_ok, NMFK_SPB_LDS);
      hipLaunchKernelGGL(sp_blk_kernel, dim3(sp_blk_grid(a, cnt)), dim3(1024), NMFK_SPB_LDS, s, a, u0, cnt);
    }
    return;
  }
#endif
  // lanes per lane element: four signals each; split: a workgroup per pass
  const dim3 grid(((a.L + NMFK_TILE - 1) / NMFK_TILE) * (a.split ? nmfk_sp_lpr(kp) : 1), a.split ? (cnt + 7) / 8 * 8 : cnt);
  const dim3 blk(NMFK_TILE);
  if (kp <= 4)
    hipLaunchKernelGGL((sp_step_kernel<1>), grid, blk, 0, s, a, u0, cnt);
  else if (kp <= 8)
    hipLaunchKernelGGL((sp_step_kernel<2>), grid, blk, 0, s, a, u0, cnt);
  else if (kp <= 16)
    hipLaunchKernelGGL((sp_step_kernel<4>), grid, blk, 0, s, a, u0, cnt);
  else if (kp <= 32)
    hipLaunchKernelGGL((sp_step_kernel<8>), grid, blk, 0, s, a, u0, cnt);
  else
    hipLaunchKernelGGL((sp_step_kernel<16>), grid, blk, 0, s, a, u0, cnt);
}

// objective of units [u0, u0 + cnt): ssepart[0] = <W'W, HH'>, ssepart[1 + tile] = non-zero terms
// parts: 1 = the non-zero terms, 2 = the Gram term (the deferred check launches only the latter: the next H half-step leaves the former)
void NMFK_NAME(nmfk_launch_sp_obj)(const void *argsv, int n, int m, int hsel, int total_iters, double weight, int u0,
                                   int cnt, hipStream_t s, int parts) {
  const NmfkSparseArgs &a = *(const NmfkSparseArgs *)argsv;  // CSR view: L = n
  if (parts & 1)
    hipLaunchKernelGGL(sp_obj_kernel, dim3((a.L + NMFK_TILE - 1) / NMFK_TILE, cnt), dim3(NMFK_TILE), 0, s, a, hsel,
                       total_iters, weight, u0);
#ifdef NMFK_IS_F32
  if ((parts & 1) && a.ell) {  // units of ranks up to 32 in the blocked form (either kernel leaves the other's units alone)
    static std::atomic<uint64_t> lds_ok{0};
    nmfk_allow_dynamic_lds((const void *)sp_blk_obj_kernel, lds_ok, NMFK_SPB_LDS);
    hipLaunchKernelGGL(sp_blk_obj_kernel, dim3(sp_blk_grid(a, cnt)), dim3(1024), NMFK_SPB_LDS, s, a, hsel, total_iters, weight,
                       u0, cnt);
  }
#endif
  if (!(parts & 2)) return;
  const int chunks = (n + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS + (m + NMFK_GRAM_ROWS - 1) / NMFK_GRAM_ROWS;
  hipLaunchKernelGGL(sp_gram_part_kernel, dim3(chunks, cnt), dim3(NMFK_TILE), 0, s, a, n, m, hsel, total_iters, u0);
  hipLaunchKernelGGL(sp_gram_dot_kernel, dim3(cnt), dim3(NMFK_TILE), 0, s, a, n, m, weight, u0);
}

void NMFK_NAME(nmfk_launch_reduce)(const NmfkStepArgs &a, int u0, int cnt, hipStream_t s) {
  hipLaunchKernelGGL(reduce_kernel, dim3(a.which == 0 ? a.PH : a.PW, cnt), dim3(NMFK_TILE), 0, s, a, u0);
}

void NMFK_NAME(nmfk_launch_sse)(const NmfkSseArgs &a, int u0, int cnt, hipStream_t s) {
  const int ntile = (a.n + NMFK_TILE - 1) / NMFK_TILE;
  if (sizeof(T) == 4 && !a.Wgt && !a.force && a.n >= 2 * NMFK_TILE)  // the loop's monitored objective, fp32 compute
    hipLaunchKernelGGL(sse_pk_kernel, dim3((ntile + 1) / 2, cnt), dim3(NMFK_TILE), 0, s, a, u0);
  else
    hipLaunchKernelGGL(sse_kernel, dim3(ntile, cnt), dim3(NMFK_TILE), 0, s, a, u0);
}

void NMFK_NAME(nmfk_launch_sum_parts)(char *arena, const NmfkRun *runs, int nunits, int ntile, double *out, hipStream_t s) {
  hipLaunchKernelGGL(sum_parts_kernel, dim3((nunits + 63) / 64), dim3(64), 0, s, arena, runs, nunits, ntile, out);
}

// parts: 1 = check_a, 2 = clamp, 4 = check_b.  The deferred check (nmfk_mu_sweep) launches the clamp in the check iteration and
// the other two behind the next H half-step, which leaves the objective.
void NMFK_NAME(nmfk_launch_check)(const NmfkCheckArgs &a, int u0, int cnt, hipStream_t s, int parts) {
  if (parts & 1) hipLaunchKernelGGL(check_a_kernel, dim3((cnt + 63) / 64), dim3(64), 0, s, a, u0, cnt);
  if (parts & 2) hipLaunchKernelGGL(clamp_kernel, dim3(a.w_clamped ? a.PH : std::max(a.PW, a.PH), cnt), dim3(NMFK_TILE), 0, s, a, u0);
  if (parts & 4) hipLaunchKernelGGL(check_b_kernel, dim3(cnt), dim3(NMFK_TILE), 0, s, a, u0);
}

void NMFK_NAME(nmfk_launch_finish)(const NmfkFinishArgs &a, hipStream_t s) {
  const int64_t work = (int64_t)std::max(a.n, a.m) * 16;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(64, work / 65536));
  hipLaunchKernelGGL(finish_kernel, dim3(nb, a.nunits), dim3(NMFK_TILE), 0, s, a);
  hipLaunchKernelGGL(finish_state_kernel, dim3((a.nunits + 63) / 64), dim3(64), 0, s, a);
}
