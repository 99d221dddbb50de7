// Split-operand MFMA half-step of libnmfk_hip for ranks 5..16 (gfx950 only; fp32 results, dense X, no missing data).
//
// The reference's half-step (src/NMFkMultiplicative.jl:67,70)  A = A .* ((X ./ (A B'))' ... ) ./ sum(B)  is, per
// lane element l and loop step d,        p = <a_l, b_d>;   q = x[l,d] / p;   num_l += q * b_d.
// The packed-VALU kernel (nmfk_step_impl.h) is bound by the vector issue port: 2k FMAs per element.  Here
//   * P = B A' runs on the matrix pipe in bf16 with BOTH operands split exactly into three bf16 terms
//     (x = x_h + x_m + x_l, 8 + 8 + 8 significand bits) and the six products of weight >= 2^-16 kept
//     (hh, hm, mh, mm, hl, lh; the dropped ml, lm, ll are below 2^-24 of |a||b|, i.e. below fp32 rounding);
//     bf16 x bf16 products are exact in the fp32 accumulator, so P has fp32 accuracy at 1/5 of the fp32-MFMA cost:
//     v_mfma_f32_16x16x32_bf16 has a contraction of 32, which holds two (k <= 16) or four (k <= 8) of the six
//     term blocks at once -> 3 or 2 MFMAs of 16 cycles per 16 x 16 tile of P;
//   * Q = X ./ P on the VALU (v_rcp_f32 + mul), it comes out of the MFMA in B-operand layout;
//   * N += B' Q on the matrix pipe in plain fp32 (v_mfma_f32_16x16x4_f32, 4 per tile), no rounding of Q.
// A factor is therefore kept in three forms: fp32 rows [L][k] (results, finish, objective), bf16 split rows
// [L+16][3][KS] (first product, as lane factor and as loop factor) and fp32 transposed [KS][ld] (second product, as
// loop factor: a lane's four loop steps are one 16-byte load).  The fused finish writes all three.
#include "nmfk_common.h"
#include "../../include/nmfk_hip.h"
#include <algorithm>
#include <type_traits>

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#ifdef HYB_DBG_NOBAR
#define HYB_BARRIER() __builtin_amdgcn_wave_barrier()
#else
#define HYB_BARRIER() __syncthreads()
#endif
__device__ __forceinline__ float hyb_div(float x, float p) { return x * __builtin_amdgcn_rcpf(p); }

// term blocks of the first product: MFMA j, k-lane group g -> split index (0 = h, 1 = m, 2 = l) of the loop factor
// (A operand) and of the lane factor (B operand; -1 = zero block)
template <int KS>
__device__ __forceinline__ int hyb_sa(int j, int g) {
  if (KS == 16) return j == 0 ? 0 : j == 1 ? 1 : (g < 2 ? 2 : 0);
  return j == 0 ? (g < 3 ? 0 : 1) : (g == 0 ? 1 : g == 1 ? 2 : 0);
}
template <int KS>
__device__ __forceinline__ int hyb_sb(int j, int g) {
  if (KS == 16) return j < 2 ? (g < 2 ? 0 : 1) : (g < 2 ? 0 : 2);
  return j == 0 ? (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 2 : 0) : (g == 0 ? 1 : g == 1 ? 0 : -1);
}

__device__ __forceinline__ uint32_t bf16_bits(float v) {  // round to nearest even (v_cvt_pk_bf16_f32)
  return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ float bf16_val(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
// v = h + m + l with three bf16 terms: the residuals v - h and (v - h) - m are exact in fp32
__device__ __forceinline__ void split3(float v, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = bf16_bits(v);
  const float r1 = v - bf16_val(h);
  m = bf16_bits(r1);
  const float r2 = r1 - bf16_val(m);
  l = bf16_bits(r2);
}

// ------------------------------------------------------------------------------------------------------
// bf16 split rows and transposed fp32 copy of a factor from its fp32 rows (after init and reduce; the fused half-step
// writes them itself and clamp_kernel patches the entries it lifts to eps).  Grid (slices, units); mask bit 0: W, bit 1: H (buffer parity hpar).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void hyb_forms_kernel(char *arena, const NmfkRun *__restrict__ runs, int n, int m,
                                                             int hpar, int mask, int u0) {
  const NmfkRun *__restrict__ rdp = runs + u0 + blockIdx.y;
  const int KS = rdp->hyb, k = rdp->k, kp = rdp->kp;
  if (KS == 0) return;
  for (int f = 0; f < 2; ++f) {
    if (!((mask >> f) & 1)) continue;
    const int L = f == 0 ? n : m, ld = f == 0 ? rdp->ldWf : rdp->ldHf;
    const float *__restrict__ F = (const float *)(arena + (f == 0 ? rdp->oWt : NMFK_HOFF(*rdp, hpar)));
    unsigned short *__restrict__ bf = (unsigned short *)(arena + (f == 0 ? rdp->oWbf : rdp->oHbf));
    float *__restrict__ ft = (float *)(arena + (f == 0 ? rdp->oWft : rdp->oHft));
    for (int l = blockIdx.x * NMFK_TILE + threadIdx.x; l < ld; l += gridDim.x * NMFK_TILE) {
      for (int c = 0; c < KS; ++c) {
        const float v = (l < L && c < k) ? F[c + (int64_t)l * kp] : 0.0f;
        ft[(int64_t)c * ld + l] = v;
        if (l < L + 16) {
          uint32_t h, mm, lo;
          split3(v, h, mm, lo);
          bf[((int64_t)l * 3 + 0) * KS + c] = (unsigned short)h;
          bf[((int64_t)l * 3 + 1) * KS + c] = (unsigned short)mm;
          bf[((int64_t)l * 3 + 2) * KS + c] = (unsigned short)lo;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// tiled copy of X for one half-step: src element (l, d) at src[d + l*D]; out block (l / 16, d / 16) = 256 floats in
// the order the MFMA layout consumes them: float index ((g*16 + c16)*4 + r) <-> l = 16 tl + c16, d = 16 td + 4g + r.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMFK_TILE) void hyb_tile_kernel(const float *__restrict__ src, int L, int D, float *__restrict__ out) {
  const int nD16 = (D + 15) >> 4, nL16 = (L + 15) >> 4;
  const int64_t total = (int64_t)nL16 * nD16 * 256;
  for (int64_t o = (int64_t)blockIdx.x * NMFK_TILE + threadIdx.x; o < total; o += (int64_t)gridDim.x * NMFK_TILE) {
    const int w = (int)(o & 255);
    const int64_t blk = o >> 8;
    const int td = (int)(blk % nD16), tl = (int)(blk / nD16);
    const int r = w & 3, c16 = (w >> 2) & 15, g = w >> 6;
    const int l = 16 * tl + c16, d = 16 * td + 4 * g + r;
    out[o] = (l < L && d < D) ? src[d + (int64_t)l * D] : 1.0f;
  }
}

// ------------------------------------------------------------------------------------------------------
// the half-step.  A wave owns NT tiles of 16 lane elements; workgroup = 4 waves with their own tiles (wsplit = 1) or
// wsplit waves sharing NT tiles and splitting the loop range.  Lane (c16 = lane & 15, g = lane >> 4).
// ------------------------------------------------------------------------------------------------------
template <int KS, int NT, int NW>  // NW: waves per workgroup when wsplit = 1 (4 or 8)
__global__ __launch_bounds__(512) void hyb_step_kernel(char *arena, const float *__restrict__ Xa,
                                                       const float *__restrict__ Xt,
                                                       const NmfkRun *__restrict__ runs,
                                                       const NmfkState *__restrict__ state,
                                                       const NmfkStepArgs *__restrict__ gp, int it, int u0) {
  extern __shared__ double lds[];  // den[16], red[8*16], cross-wave scratch
  constexpr int NM = KS == 16 ? 3 : 2;       // bf16 MFMAs of the first product
  constexpr int SUBMASK = KS == 16 ? 1 : 0;  // signal sub-block of a k-lane group: 8 * (g & SUBMASK)
  constexpr int ROWB = 3 * KS * 2;           // bytes of one bf16 split row
  const int u = u0 + blockIdx.y, bx = blockIdx.x;
  if (!gp->force && !state[u].active) return;
  const NmfkRun *__restrict__ rdp = runs + u;
  const int k = rdp->k;  // = kp: row stride of the fp32 rows and of the sum tables
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int which = gp->which, ws = gp->wsplit, S = gp->S, L = gp->L, D = gp->D;
  const int nwaves = blockDim.x >> 6;
  const int lpw = 16 * NT * (ws > 1 ? 1 : nwaves);
  const int tile = bx / S, s = bx - tile * S;
  const int l0 = tile * lpw + (ws > 1 ? 0 : wave * 16 * NT);

  const float *__restrict__ A = (const float *)(arena + (which == 0 ? NMFK_HOFF(*rdp, it) : rdp->oWt));  // lane factor
  const char *__restrict__ Abf = arena + (which == 0 ? rdp->oHbf : rdp->oWbf);
  const char *__restrict__ Bbf = arena + (which == 0 ? rdp->oWbf : rdp->oHbf);  // loop factor
  const float *__restrict__ Bft = (const float *)(arena + (which == 0 ? rdp->oWft : rdp->oHft));
  const int ldA = which == 0 ? rdp->ldHf : rdp->ldWf, ldB = which == 0 ? rdp->ldWf : rdp->ldHf;

  int d0 = s * gp->dchunk;
  int d1 = min(D, d0 + gp->dchunk);
  if (ws > 1) {
    const int q = (((d1 - d0 + ws - 1) / ws) + 15) & ~15;  // equal shares of the range per wave, in whole chunks
    d0 = min(d0 + wave * q, d1);
    d1 = min(d0 + q, d1);
  }
  d0 = __builtin_amdgcn_readfirstlane(d0);
  d1 = __builtin_amdgcn_readfirstlane(d1);

  // lane-factor operand blocks of the first product (loop invariant)
  bf16x8_t bop[NT][NM];
  int lt[NT];
  bool lv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int l = l0 + 16 * t + c16;
    lv[t] = l < L;
    lt[t] = lv[t] ? l : 0;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const int sb = hyb_sb<KS>(j, g);
      u32x4_t w = {0u, 0u, 0u, 0u};
      if (lv[t] && sb >= 0) w = *(const u32x4_t *)(Abf + ((int64_t)lt[t] * 3 + sb) * (KS * 2) + 16 * (g & SUBMASK));
      bop[t][j] = __builtin_bit_cast(bf16x8_t, w);
    }
  }
  f32x4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // Xa = the copy of X with the loop dimension contiguous: element (l, d) at d + l*D

  // X comes from the tiled copy Xt (hyb_tile_kernel): the 16 x 16 block (lane tile, chunk) is 1 KB in lane order, so
  // a wave's load is one contiguous KB (from the plain copy each lane's 16 bytes sit in a different row: 64 separate
  // L1 accesses per load, which bounded the kernel).  Byte offset of this lane's piece of chunk 0 of its tiles:
  const int nD16 = (D + 15) >> 4, nL16 = (L + 15) >> 4;
  uint32_t xoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xoff[t] = (uint32_t)(((int64_t)min((l0 >> 4) + t, nL16 - 1) * nD16 * 256 + lane * 4) * 4);
  uint32_t aoff[NM];  // byte offsets into a chunk of bf16 split rows: row c16, block (split, sub-block)
#pragma unroll
  for (int j = 0; j < NM; ++j) aoff[j] = (uint32_t)(c16 * ROWB + hyb_sa<KS>(j, g) * (KS * 2) + 16 * (g & SUBMASK));
  // rows KS..15 of the second product's A operand do not exist: those lanes re-read row KS-1 and their numerator rows
  // (signals >= KS) are never used
  const uint32_t noff = (uint32_t)(((int64_t)min(c16, KS - 1) * ldB + 4 * g) * 4);

  // full chunks: loads run one chunk ahead in two register sets.  Buffer loads: resource base = the array, per-lane
  // part = a loop-invariant 32-bit VGPR offset, chunk position = the scalar offset -> no vector address arithmetic.
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)Xt, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void *)Bbf, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void *)Bft, 0, -1, 0x00020000);
  auto load = [&](int dch, f32x4_t (&xv)[NT], u32x4_t (&av)[NM], f32x4_t &bn) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      xv[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[t], dch * 64, 0));
#pragma unroll
    for (int j = 0; j < NM; ++j) av[j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsa, aoff[j], dch * ROWB, 0));
    bn = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsn, noff, dch * 4, 0));
  };
  auto chunk = [&](int dch, const f32x4_t (&xcur)[NT], const u32x4_t (&av)[NM], const f32x4_t &bn, auto full_tag)
                   __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    f32x4_t p[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#ifdef HYB_DBG_NOM1
        p[t] += __builtin_bit_cast(f32x4_t, av[j]);
#else
        p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[j]), bop[t][j], p[t], 0, 0, 0);
#endif
      }
    // p[t][r] = <a_l, b_d> at d = dch + 4g + r, l = l0 + 16t + c16
    f32x4_t q[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#ifdef HYB_DBG_NORCP
        q[t][r] = xcur[t][r] * p[t][r];
#else
        q[t][r] = hyb_div(xcur[t][r], p[t][r]);
#endif
        if (!FULL) q[t][r] = (dch + 4 * g + r < d1) ? q[t][r] : 0.0f;
      }
#ifdef HYB_DBG_NOM2
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0] + bn[1] + bn[2] + bn[3], (q[t][0] + q[t][1]) + (q[t][2] + q[t][3]), acc[t], 0, 0, 0);
#else
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[r], q[t][r], acc[t], 0, 0, 0);
#endif
  };
  // inputs of the fused finish, fetched before the loop so that the finish does not wait for memory: the other
  // factor's sums (denominators of Mult:67 / Mult:70) and this lane's old factor values
  const bool fused = gp->fused != 0;
  double *den = lds;
  const bool vec4 = (k & 3) == 0;  // fp32 rows are 16-byte aligned: one load / store per lane instead of four
  f32x4_t aold[NT];
  if (fused) {
    const double *sumB = (const double *)(arena + (which == 0 ? rdp->osumW : rdp->osumH));
    const int PB = which == 0 ? gp->PW : gp->PH;
    if (tid < k) {
      double sd = 0;
      for (int pp = 0; pp < PB; ++pp) sd += sumB[pp * k + tid];
      den[tid] = sd;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      aold[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (vec4) {
        if (4 * g < k) aold[t] = *(const f32x4_t *)(A + 4 * g + (int64_t)lt[t] * k);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < k) aold[t][r] = A[4 * g + r + (int64_t)lt[t] * k];
      }
    }
  }
#ifdef HYB_DBG_NOLOOP
  const int nfull = 0;
#else
  const int nfull = (d1 - d0) >> 4;
#endif
  char *sbase = (char *)(lds + 9 * 16);  // staging buffers (wsplit = 1) / cross-wave scratch (wsplit > 1)
  if (ws == 1) {
    // The waves of the workgroup walk the same loop range: the loop factor's chunks are fetched ONCE per workgroup
    // into LDS (two chunks = 32 loop steps per barrier, double-buffered) and every wave reads its operand blocks
    // from there -- the vector L1 (64 B/clk/CU) only carries X and one copy of the loop factor instead of one per
    // wave, which is what bounded the first version of this kernel.
    constexpr int CPB = 4;                   // chunks per staged block (one barrier per block)
    constexpr int RS = KS == 16 ? 112 : 48;  // LDS row stride of the split rows, FRS of the transposed rows: the b128
    constexpr int FRS = 80;                  // reads of 16 lanes hit 16 different groups of 4 banks (64 banks)
    constexpr int BFB = 16 * CPB * RS, STB = BFB + CPB * 16 * FRS;  // bytes of a staged block: split rows, then [chunk][c][16 d]
    constexpr int PR = ROWB / 16, PBF = 16 * CPB * PR, NP = PBF + KS * 4 * CPB;  // 16-byte pieces of a block
    constexpr int NPT = (NP + 64 * NW - 1) / (64 * NW);
    const int64_t obf = which == 0 ? rdp->oWbf : rdp->oHbf, oft = which == 0 ? rdp->oWft : rdp->oHft;
    const int64_t ob = obf < oft ? obf : oft;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void *)(arena + ob), 0, -1, 0x00020000);
    uint32_t goff[NPT], gstep[NPT], lofs[NPT];
    bool pv[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int pc = tid + 64 * NW * i;
      pv[i] = pc < NP;
      if (pc < PBF) {
        const int row = pc / PR, part = pc - row * PR;
        goff[i] = (uint32_t)(obf - ob) + row * ROWB + part * 16;
        gstep[i] = ROWB;
        lofs[i] = row * RS + part * 16;
      } else {
        const int q = min(pc, NP - 1) - PBF, c = q / (4 * CPB), part = q - c * (4 * CPB);
        goff[i] = (uint32_t)(oft - ob) + (uint32_t)(((int64_t)c * ldB + part * 4) * 4);
        gstep[i] = 4;
        lofs[i] = BFB + ((part >> 2) * 16 + c) * FRS + (part & 3) * 16;
      }
    }
    auto stage_load = [&](int dch, u32x4_t (&sv)[NPT]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NPT; ++i)
        if (pv[i]) sv[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsb, goff[i] + dch * gstep[i], 0, 0));
    };
    auto stage_write = [&](int buf, const u32x4_t (&sv)[NPT]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NPT; ++i)
        if (pv[i]) *(u32x4_t *)(sbase + buf * STB + lofs[i]) = sv[i];
    };
    auto xload = [&](int dch, f32x4_t (&xv)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#ifdef HYB_DBG_NOX
        xv[t] = (f32x4_t){1.f, 2.f, 3.f, 4.f};
#else
        xv[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[t], dch * 64, 0));
#endif
    };
    const int fofs = c16 * RS + 16 * (g & SUBMASK), nofs = BFB + min(c16, KS - 1) * FRS + g * 16;
    auto lds_chunk = [&](int buf, int ch, int dch, const f32x4_t (&xv)[NT]) __attribute__((always_inline)) {
      const char *b = sbase + buf * STB;
      u32x4_t av[NM];
#pragma unroll
      for (int j = 0; j < NM; ++j) av[j] = *(const u32x4_t *)(b + ch * 16 * RS + fofs + hyb_sa<KS>(j, g) * (KS * 2));
      const f32x4_t bn = *(const f32x4_t *)(b + nofs + ch * 16 * FRS);
      chunk(dch, xv, av, bn, std::true_type());
    };
    int dch = d0;
    for (int i = 0; i < (nfull & (2 * CPB - 1)); ++i, dch += 16) {  // chunks ahead of the pipeline: straight from memory
      f32x4_t xv[NT], bn;
      u32x4_t av[NM];
      load(dch, xv, av, bn);
      chunk(dch, xv, av, bn, std::true_type());
    }
    const int dend = d0 + 16 * nfull;
    if (dch < dend) {
      // pipeline over pairs of blocks (2 * CPB chunks per trip): X runs two chunks ahead in four register sets, the
      // block after next is fetched while a block is computed and written to the free LDS buffer at the block's end
      f32x4_t xr[4][NT];
      u32x4_t sv[NPT];
      const int dlast = dend - 16;
      stage_load(dch, sv);
      xload(dch, xr[0]);
      xload(dch + 16, xr[1]);
      stage_write(0, sv);
      __syncthreads();
      for (; dch < dend; dch += 32 * CPB) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int db = dch + 16 * CPB * hb;
          stage_load(min(db + 16 * CPB, dend - 16 * CPB), sv);  // (past the end: the last block again, unused)
#pragma unroll
          for (int ch = 0; ch < CPB; ++ch) {
            const int ci = hb * CPB + ch;
            xload(min(db + 16 * (ch + 2), dlast), xr[(ci + 2) & 3]);
            __builtin_amdgcn_sched_barrier(0);  // loads stay in front of the arithmetic they overlap with
            lds_chunk(hb, ch, db + 16 * ch, xr[ci & 3]);
            __builtin_amdgcn_sched_barrier(0);
          }
          stage_write(hb ^ 1, sv);
          HYB_BARRIER();
        }
      }
    }
  } else {
    f32x4_t x0[NT], x1[NT], b0, b1;
    u32x4_t a0[NM], a1[NM];
    int dch = d0;
    if (nfull & 1) {  // odd count: one chunk ahead of the two-chunk pipeline
      load(dch, x0, a0, b0);
      chunk(dch, x0, a0, b0, std::true_type());
      dch += 16;
    }
    const int dend = d0 + 16 * nfull;
    if (dch < dend) {
      load(dch, x0, a0, b0);
      for (; dch < dend; dch += 32) {
        load(dch + 16, x1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);  // the loads of the next chunk stay in front of this chunk's arithmetic
        chunk(dch, x0, a0, b0, std::true_type());
        __builtin_amdgcn_sched_barrier(0);
        load(min(dch + 32, dend - 16), x0, a0, b0);  // (past the end: the last chunk again, unused)
        __builtin_amdgcn_sched_barrier(0);
        chunk(dch + 16, x1, a1, b1, std::true_type());
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if ((d1 - d0) & 15) {  // ragged end of the loop range: element-wise X loads inside the row, loop steps >= d1 masked
    const int dch = d0 + 16 * nfull;
    f32x4_t xv[NT], bn;
    u32x4_t av[NM];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[t][r] = Xa[(int64_t)lt[t] * D + min(dch + 4 * g + r, D - 1)];
    const char *ab = Bbf + (int64_t)dch * ROWB;
#pragma unroll
    for (int j = 0; j < NM; ++j) av[j] = *(const u32x4_t *)(ab + aoff[j]);
    bn = *(const f32x4_u *)((const char *)(Bft + dch) + noff);
    chunk(dch, xv, av, bn, std::false_type());
  }
  // acc[t][r] = numerator of signal c = 4g + r at lane element l0 + 16t + c16

  float *scratch = (float *)sbase;
  if (ws > 1) {  // add the waves' numerators in wave order
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[(((wave - 1) * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll 1
      for (int w = 0; w < ws - 1; ++w)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] += scratch[((w * NT + t) * 4 + r) * 64 + lane];
    }
  }
  const bool owner = (ws == 1) || (wave == 0);

  if (!fused) {
    if (owner) {
      float *__restrict__ part = (float *)(arena + rdp->opart);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (lv[t]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 4 * g + r;
            if (c < k) part[((int64_t)s * L + lt[t]) * k + c] = acc[t][r];
          }
        }
    }
    return;
  }

  __syncthreads();  // den[] is visible
  float *__restrict__ Anew = which == 0 ? (float *)(arena + NMFK_HOFF(*rdp, it + 1)) : (float *)(arena + rdp->oWt);
  char *__restrict__ Abfw = arena + (which == 0 ? rdp->oHbf : rdp->oWbf);
  float *__restrict__ Aftw = (float *)(arena + (which == 0 ? rdp->oHft : rdp->oWft));
  double *sumA = (double *)(arena + (which == 0 ? rdp->osumH : rdp->osumW)) + (int64_t)tile * k;
  double *red = den + 16;  // [8][16]
  float vs[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef HYB_DBG_NOFIN
  if (owner && acc[0][0] == 12345.f) {
#else
  if (owner) {
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (lv[t]) {
        float v[4];
        uint32_t h[4], mm[4], lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * g + r;
          v[r] = 0.f;
          if (c < k) {
            v[r] = aold[t][r] * acc[t][r] / (float)den[c];  // Mult:67 / Mult:70 order
            if (!vec4) Anew[c + (int64_t)lt[t] * k] = v[r];
          }
          if (c < KS) Aftw[(int64_t)c * ldA + lt[t]] = v[r];
          vs[r] += v[r];
          split3(v[r], h[r], mm[r], lo[r]);
        }
        if (vec4 && 4 * g < k) *(f32x4_t *)(Anew + 4 * g + (int64_t)lt[t] * k) = (f32x4_t){v[0], v[1], v[2], v[3]};
        if (4 * g < KS) {
          char *row = Abfw + (int64_t)lt[t] * ROWB + 8 * g;
          *(u32x2_t *)(row + 0 * KS * 2) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
          *(u32x2_t *)(row + 1 * KS * 2) = (u32x2_t){mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16)};
          *(u32x2_t *)(row + 2 * KS * 2) = (u32x2_t){lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
        }
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = 4 * g + r;
    double v = (double)vs[r];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (c16 == 0 && c < k) red[wave * 16 + c] = v;
  }
  __syncthreads();
  if (tid < k) {
    double t = red[tid];
    if (ws == 1)
      for (int w = 1; w < nwaves; ++w) t += red[w * 16 + tid];
    sumA[tid] = t;
  }
}

// ------------------------------------------------------------------------------------------------------
// Monitored objective (Mult:74) of the units of the split-operand MFMA group: sum(((X - W*H) * weight)^2) with W*H from
// the same three-term bf16 products as the half-step (fp32-accurate), residuals squared and accumulated in fp64.
// Dense X without missing entries, scalar weight.  Lanes = rows of X (W orientation: tiled copy Xt of the W half-step,
// loop over the columns = rows of H's split form).  Grid ((n + 255) / 256, units), 256 threads: a wave owns 4 tiles of
// 16 rows; one partial per workgroup in ossepart[] like sse_kernel.
// ------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(NMFK_TILE) void hyb_sse_kernel(char *arena, const float *__restrict__ Xt,
                                                            const NmfkRun *__restrict__ runs,
                                                            const NmfkState *__restrict__ state, int n, int m, double weight,
                                                            int u0) {
  __shared__ double sh[8];
  constexpr int NT = 4, NM = KS == 16 ? 3 : 2, SUBMASK = KS == 16 ? 1 : 0, ROWB = 3 * KS * 2;
  const int u = u0 + blockIdx.y;
  if (!state[u].active) return;
  const NmfkRun *__restrict__ rdp = runs + u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int L = n, D = m, nD16 = (D + 15) >> 4, nL16 = (L + 15) >> 4;
  const int l0 = blockIdx.x * NMFK_TILE + wave * 64;
  const char *__restrict__ Abf = arena + rdp->oWbf;  // lane factor W
  const char *__restrict__ Bbf = arena + rdp->oHbf;  // loop factor H
  bf16x8_t bop[NT][NM];
  bool lv[NT];
  uint32_t xoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int l = l0 + 16 * t + c16;
    lv[t] = l < L;
    const int lt = lv[t] ? l : 0;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const int sb = hyb_sb<KS>(j, g);
      u32x4_t w = {0u, 0u, 0u, 0u};
      if (lv[t] && sb >= 0) w = *(const u32x4_t *)(Abf + ((int64_t)lt * 3 + sb) * (KS * 2) + 16 * (g & SUBMASK));
      bop[t][j] = __builtin_bit_cast(bf16x8_t, w);
    }
    xoff[t] = (uint32_t)(((int64_t)min((l0 >> 4) + t, nL16 - 1) * nD16 * 256 + lane * 4) * 4);
  }
  uint32_t aoff[NM];
#pragma unroll
  for (int j = 0; j < NM; ++j) aoff[j] = (uint32_t)(c16 * ROWB + hyb_sa<KS>(j, g) * (KS * 2) + 16 * (g & SUBMASK));
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)Xt, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void *)Bbf, 0, -1, 0x00020000);
  double ssum = 0.0;
  for (int dch = 0; dch < D; dch += 16) {  // (the split rows are zero-padded by 16, the tiled X by whole blocks)
    f32x4_t xv[NT], p[NT];
    u32x4_t av[NM];
#pragma unroll
    for (int t = 0; t < NT; ++t)
      xv[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[t], dch * 64, 0));
#pragma unroll
    for (int j = 0; j < NM; ++j)
      av[j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsa, aoff[j], dch * ROWB, 0));
#pragma unroll
    for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[j]), bop[t][j], p[t], 0, 0, 0);
    const bool full = dch + 16 <= D;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = xv[t][r] - p[t][r];
        const double e2 = (double)e * (double)e;
        ssum += (lv[t] && (full || dch + 4 * g + r < D)) ? e2 : 0.0;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ssum += __shfl_down(ssum, o, 64);
  if (lane == 0) sh[wave] = ssum;
  __syncthreads();
  if (tid == 0) ((double *)(arena + rdp->ossepart))[blockIdx.x] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * weight * weight;
}

}  // namespace

#ifndef NMFK_HYB_NT
#define NMFK_HYB_NT 2  // 16-wide lane tiles per wave
#endif
#ifndef NMFK_HYB_NW
#define NMFK_HYB_NW 8  // waves per workgroup (wsplit = 1): they share the staged chunks of the loop factor
#endif

int nmfk_hyb_lane_tile(int wsplit) { return 16 * NMFK_HYB_NT * (wsplit > 1 ? 1 : NMFK_HYB_NW); }

// half-step of the `cnt` units [u0, u0 + cnt), all of split width ks (8: k <= 8, 16: k <= 16)
void nmfk_launch_step_hyb_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int ks, int u0, int cnt, hipStream_t s) {
  constexpr int NT = NMFK_HYB_NT, NW = NMFK_HYB_NW;
  const int ws = a.wsplit, nwaves = ws > 1 ? ws : NW;
  const int lpw = 16 * NT * (ws > 1 ? 1 : nwaves);
  const int ntile = (a.L + lpw - 1) / lpw;
  const dim3 grid(ntile * a.S, cnt), blk(64 * nwaves);
  const size_t cross = ws > 1 ? (size_t)(ws - 1) * NT * 4 * 64 * sizeof(float) : 0;
  const size_t stage = 2 * (64 * (ks == 16 ? 112 : 48) + 4 * 16 * 80);  // two staged blocks of 4 chunks (wsplit = 1)
  const size_t ldsb = sizeof(double) * 9 * 16 + std::max(cross, stage);
  if (ks == 8)
    hipLaunchKernelGGL((hyb_step_kernel<8, NT, NW>), grid, blk, ldsb, s, a.arena, a.Xalt, a.Xtile, a.runs, a.state, dargs, a.it, u0);
  else
    hipLaunchKernelGGL((hyb_step_kernel<16, NT, NW>), grid, blk, ldsb, s, a.arena, a.Xalt, a.Xtile, a.runs, a.state, dargs, a.it, u0);
}

void nmfk_launch_hyb_forms(char *arena, const NmfkRun *runs, int n, int m, int hpar, int mask, int u0, int cnt,
                           hipStream_t s) {
  const int nb = std::max(1, std::min(32, (std::max(n, m) + 16 + NMFK_TILE - 1) / NMFK_TILE));
  hipLaunchKernelGGL(hyb_forms_kernel, dim3(nb, cnt), dim3(NMFK_TILE), 0, s, arena, runs, n, m, hpar, mask, u0);
}

// tiled copy of X (element (l, d) at src[d + l*D]) for nmfk_launch_step_hyb_f32; out: roundup16(L) * roundup16(D) floats
void nmfk_launch_hyb_tile(const float *src, int L, int D, float *out, hipStream_t s) {
  const int64_t total = (int64_t)((L + 15) / 16) * ((D + 15) / 16) * 256;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (total + NMFK_TILE - 1) / NMFK_TILE));
  hipLaunchKernelGGL(hyb_tile_kernel, dim3(nb), dim3(NMFK_TILE), 0, s, src, L, D, out);
}

// monitored objective of the units [u0, u0 + cnt) of a group on the split-operand MFMA kernel (active units only)
void nmfk_launch_hyb_sse(char *arena, const float *Xtile_w, const NmfkRun *runs, const NmfkState *state, int n, int m,
                         double weight, int ks, int u0, int cnt, hipStream_t s) {
  const dim3 grid((n + NMFK_TILE - 1) / NMFK_TILE, cnt), blk(NMFK_TILE);
  if (ks == 8)
    hipLaunchKernelGGL((hyb_sse_kernel<8>), grid, blk, 0, s, arena, Xtile_w, runs, state, n, m, weight, u0);
  else
    hipLaunchKernelGGL((hyb_sse_kernel<16>), grid, blk, 0, s, arena, Xtile_w, runs, state, n, m, weight, u0);
}
