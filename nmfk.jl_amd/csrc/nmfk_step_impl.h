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

// out[c] = sum_l F[c + l*kp], c < k  (colsum(W) / rowsum(H) in the signal-major layout), fp64 accumulation
__device__ __forceinline__ void block_signal_sums(const T *F, int kp, int k, int len, T *out, double *sh) {
  for (int c = 0; c < k; ++c) {
    double s = 0;
    for (int l = threadIdx.x; l < len; l += NMFK_TILE) s += (double)F[c + (int64_t)l * kp];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[c] = (T)s;
  }
  for (int c = k + threadIdx.x; c < kp; c += NMFK_TILE) out[c] = (T)0;
}

// ------------------------------------------------------------------------------------------------------
// init: W = rand(n,k) then H = rand(k,m) (Mult:38,48) or the caller's Winit/Hinit (Mult:40-41,50-51);
// state of Mult:57-63; colsum(W), rowsum(H).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void init_kernel(NmfkInitArgs g) {
  __shared__ double sh[8];
  const int u = blockIdx.x;
  const NmfkRun rd = g.runs[u];
  const int k = rd.k, kp = rd.kp, n = g.n, m = g.m;
  T *Wt = NMFK_PTR(T, g, rd.oWt);
  T *H = NMFK_PTR(T, g, rd.oH0);
  const float *Wi = g.Winit ? g.Winit[rd.kidx] : nullptr;
  const float *Hi = g.Hinit ? g.Hinit[rd.kidx] : nullptr;
  const uint64_t key = nmfk_splitmix64(rd.seed);
  int bad = 0;
  for (int64_t e = threadIdx.x; e < (int64_t)n * kp; e += NMFK_TILE) {
    const int c = (int)(e % kp);
    const int64_t i = e / kp;
    float v = 0.f;
    if (c < k) v = Wi ? Wi[(int64_t)rd.ridx * n * k + i + (int64_t)c * n] : nmfk_uniform_keyed(key, (uint64_t)(i + (int64_t)c * n));
    bad |= (v != v);
    Wt[e] = (T)v;
  }
  for (int64_t e = threadIdx.x; e < (int64_t)m * kp; e += NMFK_TILE) {
    const int c = (int)(e % kp);
    const int64_t j = e / kp;
    float v = 0.f;
    if (c < k)
      v = Hi ? Hi[(int64_t)rd.ridx * m * k + c + j * k] : nmfk_uniform_keyed(key, (uint64_t)((int64_t)n * k + c + j * k));
    bad |= (v != v);
    H[e] = (T)v;
  }
  if (bad) atomicOr(g.nan_flag, 1);
  __syncthreads();
  block_signal_sums(Wt, kp, k, n, NMFK_PTR(T, g, rd.osumW), sh);
  block_signal_sums(H, kp, k, m, NMFK_PTR(T, g, rd.osumH), sh);
  if (threadIdx.x == 0) {
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
    s.pad = 0;
    g.state[u] = s;
  }
}

// ------------------------------------------------------------------------------------------------------
// half-step numerators (the hot kernel)
// ------------------------------------------------------------------------------------------------------
template <int KP, bool NANS>
__device__ __forceinline__ void step_body(const NmfkStepArgs &g, const NmfkRun &rd) {
  const int tile = blockIdx.x / g.S;
  const int s = blockIdx.x - tile * g.S;
  const int l = tile * NMFK_TILE + threadIdx.x;
  const bool valid = l < g.L;
  const int lc = valid ? l : 0;

  const T *__restrict__ Hcur = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it));
  const T *__restrict__ Hnew = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it + 1));
  const T *__restrict__ Wt = NMFK_PTR(const T, g, rd.oWt);
  const T *__restrict__ A = g.which == 0 ? Hcur : Wt;     // lane factor
  const T *__restrict__ B = g.which == 0 ? Wt : Hnew;     // loop factor (wave-uniform rows)
  const T *__restrict__ Bold = Hcur;                      // W half-step, missing data: H before this iteration

  T a[KP], acc[KP];
#pragma unroll
  for (int c = 0; c < KP; ++c) {
    a[c] = valid ? A[c + (int64_t)lc * KP] : (T)1;
    acc[c] = (T)0;
  }
  const float *__restrict__ xp = g.X + lc;
  const int d0 = s * g.dchunk;
  const int d1 = min(g.D, d0 + g.dchunk);

  auto one = [&](int d) __attribute__((always_inline)) {
    const T *__restrict__ b = B + (int64_t)d * KP;
    T bv[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) bv[c] = b[c];
    const float xf = xp[(int64_t)d * g.ld];
    T p = (T)0;
#pragma unroll
    for (int c = 0; c < KP; ++c) p = fma_t(a[c], bv[c], p);
    T x = (T)xf;
    if (NANS) {
      // EM imputation (Mult:72): a missing entry holds fl32(W*H) of the previous iteration's result, lambda
      // on the first iteration (Mult:20).  H half-step: that IS p.  W half-step: <w_old, h_old>.
      const bool isn = xf != xf;
      if (__any(isn)) {
        T xi;
        if (g.it == 0) {
          xi = (T)g.lambda;
        } else if (g.which == 0) {
          xi = (T)(float)p;
        } else {
          const T *__restrict__ bo = Bold + (int64_t)d * KP;
          T po = (T)0;
#pragma unroll
          for (int c = 0; c < KP; ++c) po = fma_t(a[c], bo[c], po);
          xi = (T)(float)po;
        }
        x = isn ? xi : x;
      }
    }
    const T q = x / p;
#pragma unroll
    for (int c = 0; c < KP; ++c) acc[c] = fma_t(bv[c], q, acc[c]);
  };
  constexpr int U = (KP <= 8) ? 4 : 2;
  int d = d0;
  for (; d + U <= d1; d += U) {
#pragma unroll
    for (int uu = 0; uu < U; ++uu) one(d + uu);
  }
  for (; d < d1; ++d) one(d);
  if (valid) {
    T *__restrict__ part = NMFK_PTR(T, g, rd.opart) + ((int64_t)s * g.L + l) * KP;
#pragma unroll
    for (int c = 0; c < KP; ++c) part[c] = acc[c];
  }
}

#define NMFK_STEP_CASE(KP) step_body<KP, NANS>(g, rd)
template <bool NANS, bool LARGE>
__global__ __launch_bounds__(NMFK_TILE) void step_kernel(NmfkStepArgs g, int u0) {
  const int u = u0 + blockIdx.y;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  if (LARGE) {
    switch (rd.kp) {
      case 20: NMFK_STEP_CASE(20); break;
      case 24: NMFK_STEP_CASE(24); break;
      case 28: NMFK_STEP_CASE(28); break;
      case 32: NMFK_STEP_CASE(32); break;
      case 40: NMFK_STEP_CASE(40); break;
      case 48: NMFK_STEP_CASE(48); break;
      case 56: NMFK_STEP_CASE(56); break;
      case 64: NMFK_STEP_CASE(64); break;
      default: break;
    }
  } else {
    switch (rd.kp) {
      case 1: NMFK_STEP_CASE(1); break;
      case 2: NMFK_STEP_CASE(2); break;
      case 3: NMFK_STEP_CASE(3); break;
      case 4: NMFK_STEP_CASE(4); break;
      case 5: NMFK_STEP_CASE(5); break;
      case 6: NMFK_STEP_CASE(6); break;
      case 7: NMFK_STEP_CASE(7); break;
      case 8: NMFK_STEP_CASE(8); break;
      case 9: NMFK_STEP_CASE(9); break;
      case 10: NMFK_STEP_CASE(10); break;
      case 11: NMFK_STEP_CASE(11); break;
      case 12: NMFK_STEP_CASE(12); break;
      case 13: NMFK_STEP_CASE(13); break;
      case 14: NMFK_STEP_CASE(14); break;
      case 15: NMFK_STEP_CASE(15); break;
      case 16: NMFK_STEP_CASE(16); break;
      default: break;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// half-step finish: A_new = A .* (sum of partial numerators) ./ sumB ;  sumA_new   (one workgroup per unit)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void reduce_kernel(NmfkStepArgs g) {
  __shared__ double sh[8];
  const int u = blockIdx.x;
  if (!g.force && !g.state[u].active) return;
  const NmfkRun rd = g.runs[u];
  const T *Aold;
  T *Anew;
  const T *sumB;
  T *sumA;
  if (g.which == 0) {
    Aold = NMFK_PTR(const T, g, NMFK_HOFF(rd, g.it));
    Anew = NMFK_PTR(T, g, NMFK_HOFF(rd, g.it + 1));
    sumB = NMFK_PTR(const T, g, rd.osumW);
    sumA = NMFK_PTR(T, g, rd.osumH);
  } else {
    Aold = NMFK_PTR(const T, g, rd.oWt);
    Anew = NMFK_PTR(T, g, rd.oWt);
    sumB = NMFK_PTR(const T, g, rd.osumH);
    sumA = NMFK_PTR(T, g, rd.osumW);
  }
  const int k = rd.k, kp = rd.kp;
  const T *part = NMFK_PTR(const T, g, rd.opart);
  const int64_t LK = (int64_t)g.L * kp;
  for (int64_t e = threadIdx.x; e < LK; e += NMFK_TILE) {
    const int c = (int)(e % kp);
    T num = (T)0;
    for (int s = 0; s < g.S; ++s) num += part[(int64_t)s * LK + e];
    T v = Aold[e] * num / sumB[c];  // same operation order as Mult:67,70
    if (c >= k) v = (T)0;
    Anew[e] = v;
  }
  __syncthreads();
  block_signal_sums(Anew, kp, k, g.L, sumA, sh);
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
    const T e = ((T)xf - p) * wgt;
    const double e2 = (double)e * (double)e;
    bool use = valid && (xf == xf);
    if (g.force) use = use && (e == e);  // normnan skips NaN residuals too (Help:226-228)
    ssum += use ? e2 : 0.0;
  }
  ssum = block_sum(ssum, sh);
  if (threadIdx.x == 0) NMFK_PTR(double, g, rd.ossepart)[blockIdx.x] = ssum;
}

#define NMFK_SSE_CASE(KP) sse_body<KP>(g, rd, H, sh)
__global__ __launch_bounds__(NMFK_TILE) void sse_kernel(NmfkSseArgs g) {
  __shared__ double sh[8];
  const int u = blockIdx.y;
  const NmfkState st = g.state[u];
  if (!g.force && !st.active) return;
  const NmfkRun rd = g.runs[u];
  const int sel = g.hsel >= 0 ? g.hsel : ((st.active ? g.total_iters : st.iters) & 1);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, sel));
  NMFK_DISPATCH_KP(rd.kp, NMFK_SSE_CASE)
}

// ------------------------------------------------------------------------------------------------------
// check block, every 10th iteration (Mult:73-117): objective -> tol test -> bad-iteration bookkeeping ->
// clamp at eps(Float64) -> co-clustering consistency -> loop guard (Mult:64).  One workgroup per unit.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void check_kernel(NmfkCheckArgs g) {
  __shared__ double sh[8];
  __shared__ int sh_action, sh_diff;
  __shared__ int sh_first[NMFK_MAX_K];
  const int u = blockIdx.x;
  NmfkState *st = g.state + u;
  if (!st->active) return;
  const NmfkRun rd = g.runs[u];
  const int tid = threadIdx.x, k = rd.k, kp = rd.kp, n = g.n, m = g.m;
  if (tid == 0) {
    double obj = 0;
    for (int t = 0; t < g.ntile_n; ++t) obj += NMFK_PTR(const double, g, rd.ossepart)[t];
    st->last_obj = obj;
    int action = 1;
    if (obj < g.tol) {  // Mult:75-78
      st->active = 0;
      st->reason = NMFK_STOP_TOL;
      st->iters = g.it + 1;
      action = 0;
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
    sh_action = action;
    sh_diff = 0;
  }
  if (tid < NMFK_MAX_K) sh_first[tid] = 0x7fffffff;
  __syncthreads();
  if (!sh_action) return;

  // H = max.(H, eps()); W = max.(W, eps())  (Mult:99-100; eps() is Float64 eps whatever T is; NaN stays NaN)
  T *Wt = NMFK_PTR(T, g, rd.oWt);
  T *H = NMFK_PTR(T, g, NMFK_HOFF(rd, g.it + 1));
  const T eps = (T)2.220446049250313e-16;
  for (int64_t e = tid; e < (int64_t)n * kp; e += NMFK_TILE) {
    const T v = Wt[e];
    if ((int)(e % kp) < k && v < eps) Wt[e] = eps;
  }
  for (int64_t e = tid; e < (int64_t)m * kp; e += NMFK_TILE) {
    const T v = H[e];
    if ((int)(e % kp) < k && v < eps) H[e] = eps;
  }
  __syncthreads();
  block_signal_sums(Wt, kp, k, n, NMFK_PTR(T, g, rd.osumW), sh);
  block_signal_sums(H, kp, k, m, NMFK_PTR(T, g, rd.osumH), sh);

  // index[q] = argmin(H[:,q]) (first minimum; a NaN wins, as in Julia); cons[i,j] = index[i]==index[j];
  // consdiff == 0  <=>  the partition of the columns is unchanged.  Canonical form of a partition: every
  // column labelled by the first column of its class.
  int32_t *idx = NMFK_PTR(int32_t, g, rd.opart);  // scratch: the partial-numerator buffer is idle here
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
// results stored as T = Float32 (Exec:529-531).  One workgroup per unit.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void finish_kernel(NmfkFinishArgs g) {
  __shared__ double sh[8];
  __shared__ double rs[NMFK_MAX_K];
  const int u = blockIdx.x;
  NmfkState *st = g.state + u;
  const NmfkRun rd = g.runs[u];
  const int tid = threadIdx.x, k = rd.k, kp = rd.kp, n = g.n, m = g.m;
  int iters = st->iters, reason = st->reason;
  if (st->active) {  // the host loop ran out of iterations (maxiter % 10 != 0 or no check fired)
    iters = g.total_iters;
    reason = NMFK_STOP_MAXITER;
  }
  __syncthreads();
  const T *Wt = NMFK_PTR(const T, g, rd.oWt);
  const T *H = NMFK_PTR(const T, g, NMFK_HOFF(rd, iters));
  for (int c = 0; c < k; ++c) {
    double s = 0;
    for (int j = tid; j < m; j += NMFK_TILE) s += (double)H[c + (int64_t)j * kp];
    s = block_sum(s, sh);
    if (tid == 0) rs[c] = s;
  }
  __syncthreads();
  float *Wo = g.Wout[rd.kidx] + (int64_t)rd.ridx * n * k;
  float *Ho = g.Hout[rd.kidx] + (int64_t)rd.ridx * m * k;
  for (int64_t e = tid; e < (int64_t)n * k; e += NMFK_TILE) {
    const int c = (int)(e / n);
    const int64_t i = e - (int64_t)c * n;
    T v = Wt[c + i * kp];
    if (g.normalize) v = v * (T)rs[c];
    Wo[e] = (float)v;
  }
  for (int64_t e = tid; e < (int64_t)m * k; e += NMFK_TILE) {
    const int c = (int)(e % k);
    const int64_t j = e / k;
    T v = H[c + j * kp];
    if (g.normalize) v = v / (T)rs[c];
    Ho[e] = (float)v;
  }
  if (tid == 0) {
    double obj = 0;
    for (int t = 0; t < g.ntile_n; ++t) obj += NMFK_PTR(const double, g, rd.ossepart)[t];
    g.frob[rd.kidx][rd.ridx] = (float)sqrt(obj);
    g.iters[rd.kidx][rd.ridx] = iters;
    g.reason[rd.kidx][rd.ridx] = reason;
    st->iters = iters;
    st->reason = reason;
    st->active = 0;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
void NMFK_NAME(nmfk_launch_init)(const NmfkInitArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(init_kernel, dim3(a.nunits), dim3(NMFK_TILE), 0, s, a);
}

void NMFK_NAME(nmfk_launch_step)(const NmfkStepArgs &a, hipStream_t s) {
  const int ntile = (a.L + NMFK_TILE - 1) / NMFK_TILE;
  const int nlarge = a.nlarge;
  const dim3 blk(NMFK_TILE);
  if (nlarge > 0) {
    const dim3 grid(ntile * a.S, nlarge);
    if (a.has_nan)
      hipLaunchKernelGGL((step_kernel<true, true>), grid, blk, 0, s, a, 0);
    else
      hipLaunchKernelGGL((step_kernel<false, true>), grid, blk, 0, s, a, 0);
  }
  if (a.nunits - nlarge > 0) {
    const dim3 grid(ntile * a.S, a.nunits - nlarge);
    if (a.has_nan)
      hipLaunchKernelGGL((step_kernel<true, false>), grid, blk, 0, s, a, nlarge);
    else
      hipLaunchKernelGGL((step_kernel<false, false>), grid, blk, 0, s, a, nlarge);
  }
}

void NMFK_NAME(nmfk_launch_reduce)(const NmfkStepArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(reduce_kernel, dim3(a.nunits), dim3(NMFK_TILE), 0, s, a);
}

void NMFK_NAME(nmfk_launch_sse)(const NmfkSseArgs &a, hipStream_t s) {
  const int ntile = (a.n + NMFK_TILE - 1) / NMFK_TILE;
  hipLaunchKernelGGL(sse_kernel, dim3(ntile, a.nunits), dim3(NMFK_TILE), 0, s, a);
}

void NMFK_NAME(nmfk_launch_check)(const NmfkCheckArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(check_kernel, dim3(a.nunits), dim3(NMFK_TILE), 0, s, a);
}

void NMFK_NAME(nmfk_launch_finish)(const NmfkFinishArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(finish_kernel, dim3(a.nunits), dim3(NMFK_TILE), 0, s, a);
}
