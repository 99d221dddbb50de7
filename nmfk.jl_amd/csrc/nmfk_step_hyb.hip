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
// Round 3: the rank is no longer padded to 16 on the matrix pipe.
//   * first product: the six term blocks need 6k contraction slots, v_mfma_f32_16x16x32_bf16 has 32: ONE instruction for
//     k <= 4 (KS = 4), two for k <= 8 (KS = 8), three for k <= 16 (KS = 16);
//   * second product: v_mfma_f32_4x4x1_16B_f32 -- sixteen independent 4 x 4 outer products per instruction, here "four
//     signals x four lane elements x one loop step" -- costs 8 cycles per four signals and 64 elements, i.e. 32 * ceil(k/4)
//     matrix cycles per 16 x 16 tile against 128 for the 16-signal form: NS = ceil(k/4) sets of accumulators (NS = 0
//     keeps v_mfma_f32_16x16x4_f32, which is the same work for k = 13..16).  A lane's accumulators then hold the sums
//     over ITS four loop steps of a chunk (d = 4g + r) only; the four k-lane groups are added once, after the loop.
//   scratch/small_mfma_rate.hip measures the forms with operands in registers (profiles/r03/small_mfma_rate.txt).
// A factor lives in HBM in ONE form, its fp32 rows [L][k].  The operand forms of the two products -- bf16 split rows for
// the first, a transposed fp32 block for the second -- exist only in LDS: the waves of a workgroup read 64 loop rows
// of the loop factor once, split them on the fly and lay them out for the matrix pipe (HybStage); the lane factor's
// split blocks are built in registers before the loop.  (Round 1 kept three forms of every factor in HBM and the
// fused finish wrote all of them: 470 MB per launch of 256 factorizations, more than the half-step's X traffic.)
#include "nmfk_common.h"
#include "../../include/nmfk_hip.h"
#include <algorithm>
#include <type_traits>

#ifndef NMFK_HYB_ABL
#define NMFK_HYB_ABL 0  // measurement builds only: resident half-step with bit 0 no X loads in the chunk loop, bit 1 no LDS operand reads in it (wrong results)
#endif
#ifndef NMFK_WIDE_XA
#define NMFK_WIDE_XA 2  // chunks the X loads of wide2_step_kernel run ahead (2 or 3; 3: k = 64 2.95 -> 2.93 ms per iteration, k = 32 1.77 -> 1.98: registers)
#endif
#ifndef NMFK_WIDE_ABL
#define NMFK_WIDE_ABL 0  // measurement builds only (scripts/r6_wide_ablate.sh): bit 0 no staging writes behind the first block, 1 no X loads in the loop,
#endif                   // 2 no ratio pieces, 3 no second product, 4 no LDS operand reads in the loop, 5 X loads from the tile's first four chunks only (wrong results, same control flow)
#ifndef NMFK_HYB_CPB
#define NMFK_HYB_CPB 4  // chunks of 16 loop steps per staged block of a workgroup (one barrier per block)
#endif

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define HYB_BARRIER() __syncthreads()
__device__ __forceinline__ float hyb_div(float x, float p) { return x * __builtin_amdgcn_rcpf(p); }
#define HYB_EPS 2.220446049250313e-16f  // eps(Float64) as the clamp of Mult:99-100 compares fp32 values with it

// term blocks of the first product: MFMA j, k-lane group g -> split index (0 = h, 1 = m, 2 = l) of the loop factor
// (A operand) and of the lane factor (B operand; -1 = zero block)
// (KS = 4: ONE MFMA holds all six blocks, two per k-lane group -- A terms (h|h), (m|m), (h|l), B terms (h|m), (h|m), (l|h),
//  group 3 is a zero block; its A planes are laid out as those pairs, see HybStage, so hyb_sa is the plane index)
template <int KS>
__device__ __forceinline__ int hyb_sa(int j, int g) {
  if (KS == 4) return g < 2 ? g : 2;
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

// the same split for a PAIR of values with packed conversions: words of two bf16 terms each (first value in the low
// half), 9 vector instructions instead of 20 (v_cvt_pk_bf16_f32 rounds and packs both, the residuals are one v_pk_add)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack2(f32x2_t v) {
  const bf16x2_t r = {(__bf16)v.x, (__bf16)v.y};
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ f32x2_t bf16_unpack2(uint32_t w) {
  return (f32x2_t){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
}
__device__ __forceinline__ void split3_pair(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
  const f32x2_t v = {v0, v1};
  h = bf16_pack2(v);
  const f32x2_t r1 = v - bf16_unpack2(h);
  m = bf16_pack2(r1);
  const f32x2_t r2 = r1 - bf16_unpack2(m);
  l = bf16_pack2(r2);
}

// ---- numerators on the bf16 pipe (round 6; wide ranks only) ----------------------------------------------------------
// The second product N += B' Q as v_mfma_f32_16x16x32_bf16 (three per block of 16 signals, lane tile and chunk: 51 matrix cycles)
// instead of four v_mfma_f32_16x16x4_f32 (128, and an fp32 matrix instruction holds the SIMD's vector issue port for all of its 32
// cycles): the ratios q are split into three bf16 terms like the factors -- by TRUNCATION, q = q_h + q_m + q_l exactly (8 + 8 + 8
// significand bits), with plain vector instructions only (v_and_b32 / v_sub_f32 for the residuals, v_perm_b32 to pack: the upper
// halves of (q0, q1) ARE the truncated terms; a packed conversion or v_dot2c_f32_bf16 costs 12 cycles beside a bf16 matrix
// instruction, and the latter is not exact: profiles/r06/bf16_numerators_probe.txt).  The six products of weight >= 2^-16 are kept
// like in the first product (b_h q_h, b_m q_h, b_l q_h, b_h q_m, b_m q_m, b_h q_l; dropped: below 2^-24 of |b||q|, i.e. below fp32
// rounding).  A lane of the first product's output holds the ratios of loop steps 4g .. 4g + 3 of ITS lane element, which is one
// half of an 8 x bf16 K-slice of the B operand when the contraction slots are ordered (k-lane group g: [term x of steps 4g..4g+3 |
// term y of the same steps]) -- no cross-lane movement; the loop factor's planes in LDS are laid out to match.
//   MFMA 0: A = (b_h | b_m), B = (q_h | q_h)      MFMA 1: A = (b_h | b_m), B = (q_m | q_m)      MFMA 2: A = (b_l | b_h), B = (q_h | q_l)
// -- the duplicated halves sit on the RATIOS' side, whose operand words are built once per lane tile and chunk and serve every block
// of 16 signals; the loop factor's two operand forms (b_l | b_h) and (b_h | b_m) are read ready-made from LDS (16 bytes per lane each).
// The split costs 7.5 plain instructions per ratio: it pays where a ratio meets 32 signals or more (wide2_step_kernel), not at
// k <= 16 (the probe: - 1 %).
__device__ __forceinline__ uint32_t hyb_pack_hi(float a, float b) {  // (upper half of a, upper half of b): one v_perm_b32
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x07060302u);
}
__device__ __forceinline__ float hyb_trunc_rest(float v) {  // v - (v truncated to bf16): exact
  return v - __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xffff0000u);
}
struct HybQ {  // B operands of a lane tile's four ratios (steps 4g .. 4g + 3 of the chunk; low half of a word = the even step's term)
  bf16x8_t hh, mm, hl;
};
// bytes of a block of 16 signals x 16 loop steps in the two operand forms: plane (b_l | b_h), plane (b_h | b_m), each [k-lane group g][signal]
// 16 bytes (+ 16: the blocks' rows start on different banks for the staging writes)
constexpr int HYB_BN_BLK = 2048 + 16;

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
// Staging of the loop factor.  A block = CPB chunks of 16 loop rows.  The GT threads of a staging group (the whole
// workgroup when its waves walk the same loop range, a single wave otherwise) read the block's fp32 rows from
// memory -- item = (row, pair of adjacent signals), NI items per thread, coalesced because the rows of a factor are
// contiguous -- split every value into its three bf16 terms and write the two operand forms into an LDS buffer:
//   [0, BFB)      split rows   chunk ch at ch*CHP: PLANES of 256 B = (term t, half hf of the signals), row r of the
//                              chunk at r*16 inside a plane: its 8 signals [8 hf, 8 hf + 8) as bf16   (A operand, 1st product)
//   [BFB, STB)    transposed   chunk ch at ch*CHT: planes of 256 B = loop steps [4g, 4g + 4), signal c at c*16: four
//                              fp32 values                                                          (A operand, 2nd product)
// Planes: a ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous -- {0-3, 12-15, 20-27}, {4-11,
// 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) -- i.e. 8 rows of one k-lane group and the other 8 rows of the next.
// Every lane reads "its row (lane & 15) of the plane its k-lane group needs", so the 16 lanes of a group always hit
// the 16 different 16-byte slots of a 256-byte bank row, whatever planes they read.  (Round 1's padded row strides
// were laid out for contiguous groups of 16 lanes: 2-way conflicts on most reads, SQ_LDS_BANK_CONFLICT as large as the
// conflict-free LDS time.)
// Rows at or beyond the factor's last row are staged as zeros; the caller masks the ratios of the loop steps beyond
// its own range.
// ------------------------------------------------------------------------------------------------------
// TM: layout of the second product's operand block: 0 = none (objective), 1 = for v_mfma_f32_16x16x4_f32 (planes of 256 B),
// 2 = for v_mfma_f32_4x4x1_16B_f32: the same planes (loop steps [4g, 4g + 4) of signal c at c*16) with a plane stride of
// 320 B -- a lane reads signal 4s + (c16 & 3) of its plane g, so the 16 lanes of a read group (8 of plane g, 8 of plane
// g + 1) touch 64 B per plane, and the 64-byte skew puts the two windows on different banks.
// KS = 4: a chunk's split planes are the three operand PAIRS (h|h), (m|m), (h|l) of the single first-product MFMA: row r
// of a plane = 16 B = two blocks of four bf16 signals.
template <int KS, int CPB, int GT, int TM>
struct HybStage {
  static constexpr bool WITH_T = TM != 0;
  static constexpr int NH = KS == 4 ? 1 : KS / 8;  // halves of 8 signals (KS = 4: one pair plane per term slot)
  static constexpr int TPS = TM == 2 ? 320 : 256;  // plane stride of the transposed block
  static constexpr int CHP = 3 * NH * 256, CHT = 4 * TPS;  // bytes of a chunk's split planes / transposed planes
  static constexpr int BFB = CPB * CHP, STB = BFB + (WITH_T ? CPB * CHT : 0);
  static constexpr int PPR = KS / 2;                   // signal pairs per row
  static constexpr int NITEM = 16 * CPB * PPR, NI = (NITEM + GT - 1) / GT;
  uint32_t voff[NI], lsp[NI], ltr[NI];
  int rr[NI];
  bool pv[NI], c0[NI], c1[NI];
  __amdgpu_buffer_rsrc_t rs;
  int rowbytes, dlim;

  // F: fp32 rows of the loop factor (row stride k floats), D rows; gt: thread index inside the staging group
  __device__ __forceinline__ void init(const float *F, int k, int D, int gt) {
    rs = __builtin_amdgcn_make_buffer_rsrc((void *)F, 0, -1, 0x00020000);
    rowbytes = k * 4;
    dlim = D;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = gt + GT * i;
      pv[i] = q < NITEM;
      const int qq = pv[i] ? q : 0, r = qq / PPR, cp = qq - r * PPR;
      rr[i] = r;
      voff[i] = (uint32_t)((r * k + 2 * cp) * 4);
      c0[i] = 2 * cp < k;
      c1[i] = 2 * cp + 1 < k;
      if (KS == 4)
        lsp[i] = (uint32_t)((r >> 4) * CHP + (r & 15) * 16 + cp * 4);  // pair plane p: + p * 256, second block of the pair: + 8
      else
        lsp[i] = (uint32_t)((r >> 4) * CHP + (cp >> 2) * 256 + (r & 15) * 16 + (cp & 3) * 4);  // term t: + t * NH * 256
      ltr[i] = (uint32_t)(BFB + (r >> 4) * CHT + ((r & 15) >> 2) * TPS + 2 * cp * 16 + (r & 3) * 4);
    }
  }
  // fetch the block that starts at loop row `row0` (a block reads at most 64 rows = 4 KB past the end of the array,
  // inside the arena).  Nothing here may USE the loaded values: the loads stay in flight until write() (a select
  // right behind the load made the compiler wait for vmcnt(0) on the spot, X prefetches included).
  __device__ __forceinline__ void load(int row0, float (&v)[NI][2], int &vrow0) const {
    vrow0 = row0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {  // (threads without an item load item 0 again and drop it)
      v[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], row0 * rowbytes, 0));
      v[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[i] + 4, row0 * rowbytes, 0));
    }
  }
  // rows past the factor's end and padding signals become zeros
  __device__ __forceinline__ void write(char *dst, const float (&vin)[NI][2], int vrow0) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (NITEM % GT != 0 && !pv[i]) continue;
      const bool ok = vrow0 + rr[i] < dlim;
      const float v[1][2] = {{(ok && c0[i]) ? vin[i][0] : 0.0f, (ok && c1[i]) ? vin[i][1] : 0.0f}};
      uint32_t h, m, l;
      split3_pair(v[0][0], v[0][1], h, m, l);
      if (KS == 4) {
        *(uint32_t *)(dst + lsp[i]) = h;            // (h|h)
        *(uint32_t *)(dst + lsp[i] + 8) = h;
        *(uint32_t *)(dst + lsp[i] + 256) = m;      // (m|m)
        *(uint32_t *)(dst + lsp[i] + 256 + 8) = m;
        *(uint32_t *)(dst + lsp[i] + 512) = h;      // (h|l)
        *(uint32_t *)(dst + lsp[i] + 512 + 8) = l;
      } else {
        *(uint32_t *)(dst + lsp[i]) = h;
        *(uint32_t *)(dst + lsp[i] + NH * 256) = m;
        *(uint32_t *)(dst + lsp[i] + 2 * NH * 256) = l;
      }
      if (WITH_T) {
        *(float *)(dst + ltr[i]) = v[0][0];
        *(float *)(dst + ltr[i] + 16) = v[0][1];
      }
    }
  }
};

// split blocks of the lane factor (B operand of the first product): 8 signals [8 * (g & SUBMASK), +8) of row `l` of
// the fp32 rows A (row stride k), term hyb_sb(j, g) for MFMA j; rows that do not exist give zero blocks
// The row's signals come with one or two 16-byte loads (dword aligned: a row starts at l*k floats) whatever k is -- a row
// shorter than the split width reads into the following rows, inside the arena, and is masked -- instead of one guarded
// load per signal (sixteen branches per pair of lane tiles: a third of the W half-step's instructions outside its loop).
struct HybRows {
  f32x4_t r0, r1;  // signals [s0, s0 + 4), [s0 + 4, s0 + 8) of a lane factor row (KS = 4: r0 only)
};
template <int KS>
__device__ __forceinline__ HybRows hyb_lane_rows(const float *__restrict__ A, int k, int l, int g) {
  constexpr int SUBMASK = KS == 16 ? 1 : 0;
  const float *rp = A + (int64_t)l * k + 8 * (g & SUBMASK);
  HybRows R;
  R.r0 = *(const f32x4_u *)rp;
  R.r1 = KS == 4 ? R.r0 : (f32x4_t)(*(const f32x4_u *)(rp + 4));
  return R;
}
// `full`: wave-uniform promise that every lane's row exists and has KS signals (k = KS): no masking at all.
template <int KS, int NM>
__device__ __forceinline__ void hyb_lane_blocks_from(const HybRows &R, int k, bool valid, int g, bf16x8_t (&bop)[NM], bool full = false) {
  if (KS == 4) {  // one MFMA: the lane factor's term PAIR of k-lane group g: (h|m), (h|m), (l|h), zero
    uint32_t hh[2], mm[2], ll[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) v[e] = (full || (valid && 2 * i + e < k)) ? R.r0[2 * i + e] : 0.0f;
      split3_pair(v[0], v[1], hh[i], mm[i], ll[i]);
    }
    u32x4_t w = {0u, 0u, 0u, 0u};
    if (g < 2) w = (u32x4_t){hh[0], hh[1], mm[0], mm[1]};
    if (g == 2) w = (u32x4_t){ll[0], ll[1], hh[0], hh[1]};
    bop[0] = __builtin_bit_cast(bf16x8_t, w);
    return;
  }
  constexpr int SUBMASK = KS == 16 ? 1 : 0;
  const int s0 = 8 * (g & SUBMASK);
  uint32_t hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 2 * i + e;
      v[e] = (full || (valid && s0 + c < k)) ? (c < 4 ? R.r0[c & 3] : R.r1[c & 3]) : 0.0f;
    }
    split3_pair(v[0], v[1], hh[i], mm[i], ll[i]);
  }
  if (KS == 16) {
    // hyb_sb<16>: MFMA 0 and 1 take the same block (h for the k-lane groups 0, 1; m for 2, 3), MFMA 2 takes (h; l):
    // two selects per word instead of a chain per MFMA
    const bool lo = g < 2;
    u32x4_t w01, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w01[i] = lo ? hh[i] : mm[i];
      w2[i] = lo ? hh[i] : ll[i];
    }
    bop[0] = __builtin_bit_cast(bf16x8_t, w01);
    bop[1 % NM] = bop[0];
    bop[2 % NM] = __builtin_bit_cast(bf16x8_t, w2);
    return;
  }
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    const int sb = hyb_sb<KS>(j, g);
    u32x4_t w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = sb == 0 ? hh[i] : sb == 1 ? mm[i] : sb == 2 ? ll[i] : 0u;
    bop[j] = __builtin_bit_cast(bf16x8_t, w);
  }
}
template <int KS, int NM>
__device__ __forceinline__ void hyb_lane_blocks(const float *__restrict__ A, int k, int l, bool valid, int g,
                                                bf16x8_t (&bop)[NM]) {
  hyb_lane_blocks_from<KS, NM>(hyb_lane_rows<KS>(A, k, l, g), k, valid, g, bop);
}

// ------------------------------------------------------------------------------------------------------
// the half-step.  A wave owns NT tiles of 16 lane elements; workgroup = NW waves with their own tiles (wsplit = 1) that
// walk the same loop range and share the staged blocks, or wsplit waves sharing NT tiles and splitting the loop range
// (each wave stages its own chunks).  Lane (c16 = lane & 15, g = lane >> 4).
// ------------------------------------------------------------------------------------------------------
// OBJ: the same walk computes the monitored objective (Mult:74) instead of a half-step: sum(((X - W*H) * weight)^2) with
// W*H from the same three-term bf16 products (fp32-accurate), residuals squared and accumulated in fp64; W orientation
// (gp = the W half-step's arguments, lanes = rows of X), whole loop range per workgroup, H of iteration parity `it`;
// one partial per workgroup in ossepart[] like sse_kernel (256 rows per workgroup = its tile).
// SSE (round 4): the half-step ALSO leaves the objective of the factors it reads -- its first product is W*H of exactly those
// factors, so the residuals cost four packed instructions per lane tile and chunk instead of a launch of their own.  The H
// half-step that follows a check iteration runs in this mode (nmfk_mu_sweep, "deferred check"); one partial per workgroup
// in ossepart[blockIdx.x] (gridDim.x partials per unit: lane tiles x splits of the loop range).
// NS: sets of four signals whose numerators run as v_mfma_f32_4x4x1_16B_f32 (ceil(k / 4)); 0 = v_mfma_f32_16x16x4_f32
template <int KS, int NS, int NT, int NW, bool OBJ, bool SSE, bool LAGK>  // NW: waves per workgroup when wsplit = 1 (4 or 8); LAGK: see LAG
__device__ __forceinline__ void hyb_step_body(char *arena, const float *__restrict__ Xa, const float *__restrict__ Xt,
                                              const NmfkRun *__restrict__ runs, const NmfkState *__restrict__ state,
                                              const NmfkStepArgs *__restrict__ gp, int it, int u0, double weight,
                                              double *lds) {  // lds: den[16], red[16*16], ssep[16], staging buffers / cross-wave scratch
  static_assert(!(OBJ && SSE), "objective only, or half-step with the objective");
  constexpr int NM = KS == 16 ? 3 : KS == 8 ? 2 : 1;  // bf16 MFMAs of the first product
  constexpr int NSA = NS > 0 ? NS : 1;       // accumulator tiles per lane tile
  constexpr int TM = OBJ ? 0 : (NS > 0 ? 2 : 1);  // layout of the second product's operand block (HybStage)
  constexpr int SUBMASK = KS == 16 ? 1 : 0;  // signal sub-block of a k-lane group: 8 * (g & SUBMASK)
  const int u = u0 + blockIdx.y, bx = blockIdx.x;
  if (!(gp->force && !OBJ) && !state[u].active) return;
  const NmfkRun *__restrict__ rdp = runs + u;
  const int k = rdp->k;  // = kp: row stride of the fp32 rows and of the sum tables
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int which = gp->which, ws = OBJ ? 1 : gp->wsplit, S = OBJ ? 1 : gp->S, L = gp->L, D = gp->D;
  const int nwaves = blockDim.x >> 6;
  const int lpw = 16 * NT * (ws > 1 ? 1 : nwaves);
  const int tile = bx / S, s = bx - tile * S;
  const int l0 = tile * lpw + (ws > 1 ? 0 : wave * 16 * NT);

  const float *__restrict__ A = (const float *)(arena + (which == 0 ? NMFK_HOFF(*rdp, it) : rdp->oWt));  // lane factor
  const float *__restrict__ B =
      (const float *)(arena + (which == 0 ? rdp->oWt : NMFK_HOFF(*rdp, OBJ ? it : it + 1)));  // loop factor

  int d0 = OBJ ? 0 : s * gp->dchunk;
  int d1 = OBJ ? D : min(D, d0 + gp->dchunk);
  if (ws > 1) {
    const int q = (((d1 - d0 + ws - 1) / ws) + 15) & ~15;  // equal shares of the range per wave, in whole chunks
    d0 = min(d0 + wave * q, d1);
    d1 = min(d0 + q, d1);
  }
  d0 = __builtin_amdgcn_readfirstlane(d0);
  d1 = __builtin_amdgcn_readfirstlane(d1);

  // lane-factor operand blocks of the first product (loop invariant), built from the fp32 rows
  bf16x8_t bop[NT][NM];
  int lt[NT];
  bool lv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int l = l0 + 16 * t + c16;
    lv[t] = l < L;
    lt[t] = lv[t] ? l : 0;
    hyb_lane_blocks<KS, NM>(A, k, lt[t], lv[t], g, bop[t]);
  }
  f32x4_t accs[NT][NSA];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int sn = 0; sn < NSA; ++sn) accs[t][sn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  double ssum = 0.0;

  // X comes from the tiled copy Xt (hyb_tile_kernel): the 16 x 16 block (lane tile, chunk) is 1 KB in lane order, so
  // a wave's load is one contiguous KB.  Byte offset of this lane's piece of chunk 0 of its tiles:
  const int nD16 = (D + 15) >> 4, nL16 = (L + 15) >> 4;
  uint32_t xoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xoff[t] = (uint32_t)(((int64_t)min((l0 >> 4) + t, nL16 - 1) * nD16 * 256 + lane * 4) * 4);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)Xt, 0, -1, 0x00020000);
  auto xload = [&](int dch, f32x4_t (&xv)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      xv[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[t], dch * 64, 0));
  };
  // one chunk of 16 loop steps from the staged operands (av: split rows of the chunk, bn: its transposed block);
  // MASK: the chunk crosses the end of the loop range, ratios of the steps >= d1 are dropped
  // first product of a chunk: p[t][r] = <a_l, b_d> at d = dch + 4g + r, l = l0 + 16t + c16
  auto chunk_p = [&](const u32x4_t (&av)[NM], f32x4_t (&p)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[j]), bop[t][j], p[t], 0, 0, 0);
  };
  // ratios and second product (or the objective's residuals)
  auto chunk_n = [&](int dch, const f32x4_t (&xcur)[NT], const f32x4_t (&p)[NT], const f32x4_t (&bn)[NSA], bool mask)
                     __attribute__((always_inline)) {
    if (OBJ) {
      // The squares of a lane's 4 * NT residuals of the chunk are added in fp32 (packed), the chunk's partial then enters
      // the fp64 sum: a partial of 8 squares carries <= 5e-7 of relative rounding noise, unbiased -- over the 8192
      // partials a lane group of a 8192 x 512 X contributes that is ~1e-8 of the objective, the stop rule's tolOF = 1e-3
      // on ~3.5e5 is 3e-9 .. per-element fp64 (round 2) cost three fp64 instructions per element: the objective launch
      // of the bench sweep took as long as a half-step (702 us per check, profiles/r03/trace_micro_before.txt).
      float part = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x2_t s2 = {0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2_t e2 = (f32x2_t){xcur[t][r], xcur[t][r + 1]} - (f32x2_t){p[t][r], p[t][r + 1]};
          if (mask) {
            e2.x = dch + 4 * g + r < d1 ? e2.x : 0.0f;
            e2.y = dch + 4 * g + r + 1 < d1 ? e2.y : 0.0f;
          }
          s2 = __builtin_elementwise_fma(e2, e2, s2);
        }
        part += lv[t] ? s2.x + s2.y : 0.0f;
      }
      ssum += (double)part;
      return;
    }
    f32x4_t q[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {  // (pairs: one v_pk_mul_f32 for two ratios)
        const f32x2_t rc = {__builtin_amdgcn_rcpf(p[t][r]), __builtin_amdgcn_rcpf(p[t][r + 1])};
        const f32x2_t q2 = (f32x2_t){xcur[t][r], xcur[t][r + 1]} * rc;
        q[t][r] = q2.x;
        q[t][r + 1] = q2.y;
        if (mask) {
          q[t][r] = (dch + 4 * g + r < d1) ? q[t][r] : 0.0f;
          q[t][r + 1] = (dch + 4 * g + r + 1 < d1) ? q[t][r + 1] : 0.0f;
        }
      }
    if (NS > 0) {
      // block (g, c16 >> 2) of the instruction: signals 4 sn + (lane & 3) of loop step 4g + r (A operand) x the ratios of
      // the block's four lane elements (B operand) -> accs[t][sn][i] += b[4g + r][4 sn + i] * q[4g + r][lane element]
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int sn = 0; sn < NSA; ++sn)
#pragma unroll
          for (int t = 0; t < NT; ++t) accs[t][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bn[sn][r], q[t][r], accs[t][sn], 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) accs[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0][r], q[t][r], accs[t][0], 0, 0, 0);
    }
  };
  // the same work per lane tile, for the skewed order of the half-step's chunk (see the trip): first product of tile t,
  // its ratios, its second product
  auto p_tile = [&](int t, const u32x4_t (&av)[NM], f32x4_t (&p)[NT]) __attribute__((always_inline)) {
    p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NM; ++j)
      p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[j]), bop[t][j], p[t], 0, 0, 0);
  };
  float spart = 0.0f;  // SSE: squares of the chunk's residuals (fp32; see the OBJ mode for the error budget)
  auto q_tile = [&](int t, int dch, const f32x4_t (&xcur)[NT], const f32x4_t (&p)[NT], f32x4_t (&q)[NT], bool mask)
                    __attribute__((always_inline)) {
    if (SSE) {  // (first: the residuals need x and p, the ratios below retire both)
      float sqs = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = xcur[t][r] - p[t][r];
        if (mask) e = dch + 4 * g + r < d1 ? e : 0.0f;
        sqs = __builtin_fmaf(e, e, sqs);
      }
      spart += lv[t] ? sqs : 0.0f;
      // (anchor: without it the instruction selector's linear order puts the residuals of a whole trip next to their only
      //  consumer, the sum behind the trip's last chunk, and every chunk's x and p stay alive until then -- 300 B of scratch)
      asm volatile("" : "+v"(spart));
    }
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2_t rc = {__builtin_amdgcn_rcpf(p[t][r]), __builtin_amdgcn_rcpf(p[t][r + 1])};
      const f32x2_t q2 = (f32x2_t){xcur[t][r], xcur[t][r + 1]} * rc;
      q[t][r] = q2.x;
      q[t][r + 1] = q2.y;
      if (mask) {
        q[t][r] = (dch + 4 * g + r < d1) ? q[t][r] : 0.0f;
        q[t][r + 1] = (dch + 4 * g + r + 1 < d1) ? q[t][r + 1] : 0.0f;
      }
    }
  };
  // the same in two parts (round 5; see hyb_res_body's chunk pipeline): the reciprocals -- they go beside the bf16 matrix instructions
  // of the next lane tile's first product --, and the packed multiplies -- 12 cycles each THERE, 5.5 beside the fp32 matrix
  // instructions of the second product
  auto r_tile = [&](int t, int dch, const f32x4_t (&xcur)[NT], const f32x4_t (&p)[NT], f32x4_t (&q)[NT], bool mask)
                    __attribute__((always_inline)) {
    if (SSE) {
      float sqs = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = xcur[t][r] - p[t][r];
        if (mask) e = dch + 4 * g + r < d1 ? e : 0.0f;
        sqs = __builtin_fmaf(e, e, sqs);
      }
      spart += lv[t] ? sqs : 0.0f;
      asm volatile("" : "+v"(spart));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) q[t][r] = __builtin_amdgcn_rcpf(p[t][r]);
  };
  auto m_tile = [&](int t, int dch, const f32x4_t (&xcur)[NT], f32x4_t (&q)[NT], bool mask) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2_t q2 = (f32x2_t){xcur[t][r], xcur[t][r + 1]} * (f32x2_t){q[t][r], q[t][r + 1]};
      q[t][r] = q2.x;
      q[t][r + 1] = q2.y;
      if (mask) {
        q[t][r] = (dch + 4 * g + r < d1) ? q[t][r] : 0.0f;
        q[t][r + 1] = (dch + 4 * g + r + 1 < d1) ? q[t][r + 1] : 0.0f;
      }
    }
  };
  auto sse_chunk = [&]() __attribute__((always_inline)) {  // the chunk's partial enters the fp64 sum
    ssum += (double)spart;
    spart = 0.0f;
  };
  auto f_tile = [&](int t, const f32x4_t (&bn)[NSA], const f32x4_t (&q)[NT]) __attribute__((always_inline)) {
    if (NS > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int sn = 0; sn < NSA; ++sn) accs[t][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bn[sn][r], q[t][r], accs[t][sn], 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) accs[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0][r], q[t][r], accs[t][0], 0, 0, 0);
    }
  };
  // inputs of the fused finish, fetched before the loop so that the finish does not wait for memory: the other
  // factor's sums (denominators of Mult:67 / Mult:70) and this lane's old factor values
  const bool fused = !OBJ && gp->fused != 0;
  double *den = lds;
  const bool vec4 = (k & 3) == 0;  // fp32 rows are 16-byte aligned: one load / store per lane instead of four
  f32x4_t aold[NT];
  auto load_aold = [&]() __attribute__((always_inline)) {
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
  };
  // LAG (round 5): the second lane tile's ratios and second product run ONE CHUNK LATE, so that its reciprocals sit beside the bf16 matrix
  // instructions of the next chunk's first product (2-6 cycles each) instead of beside the fp32 ones of its own second product (8 each:
  // profiles/r05/issue_rates.txt).  Carries W*H of the tile and the second product's operand block across the chunk boundary (8
  // registers: the old factor values of the finish are fetched behind the loop instead, like in the 4x4x1 forms).
  // LAGK = false: launches whose waves walk a short loop range (one unit alone, BASELINE configs[1]: the lag's dummy in front and its tail
  // behind cost a chunk -- 44.9 -> 48.0 us per iteration with the lag everywhere)
  constexpr bool LAG = LAGK && NT == 2 && !OBJ;
  constexpr bool AOLD_EARLY = NS == 0 && !LAG;  // (the 4x4x1 forms carry more accumulators: their old values are fetched behind the loop)
  if (fused) {
    const double *sumB = (const double *)(arena + (which == 0 ? rdp->osumW : rdp->osumH));
    const int PB = which == 0 ? rdp->nsW : rdp->nsH;  // (the slots behind the unit's own are zero)
    if (tid < k) den[tid] = nmfk_slot_sum<4>(sumB, k, PB, tid);
    if (AOLD_EARLY) load_aold();
  }
  bool aold_loaded = false;  // (wave-uniform) run() fetched them in front of its last trips
  char *sbase = (char *)(lds + 18 * 16);  // staging buffers, later the cross-wave scratch (wsplit > 1)
  const int nchunks = (d1 - d0 + 15) >> 4;

  // The loop.  TRIP chunks per trip of the (unrolled) body so that the LDS buffer of a block and the register set of
  // a chunk's X entries are compile-time constants; X runs two chunks ahead in four register sets, the next block of
  // the loop factor is fetched while a block is computed and written to the free LDS buffer at the block's end.
  // (Round 4: the blocks ready-made from an operand IMAGE of the factor in HBM, staged by LDS-DMA with no registers, no
  //  conversion and no LDS writes -- built, parity-green, and exactly as fast: H half-step of the bench sweep 0.684 vs 0.682 ms,
  //  profiles/r04/operand_image_dma.txt.  The staging INSTRUCTIONS are not what the half-step waits for.)
  auto run = [&](auto stage, char *sb, auto barrier) __attribute__((always_inline)) {
    typedef decltype(stage) ST;
    constexpr int CPB = (ST::NITEM / (16 * ST::PPR));
    constexpr int TRIP = 2 * CPB > 4 ? 2 * CPB : 4;
    // this lane's row of the planes it reads: MFMA j of the first product wants term hyb_sa(j, g), half g & SUBMASK
    int fofs[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) fofs[j] = (hyb_sa<KS>(j, g) * ST::NH + (g & SUBMASK)) * 256 + c16 * 16;
    // second product's operand: 16x16x4 form: signal c16 of plane g; 4x4x1 form: signal 4 sn + (c16 & 3) (sn: + 64 B)
    const int nofs = ST::BFB + g * ST::TPS + (NS > 0 ? (c16 & 3) : min(c16, KS - 1)) * 16;
    if (nchunks <= 0) return;
    // staging registers of TWO blocks: the rows of block b + 2 are requested at the start of block b and written to LDS
    // during block b + 1 -- a whole block of latency even when a block is a single chunk (per-wave staging), where a
    // request made and consumed within one chunk exposed the latency of the loop factor's rows (they come from beyond L2)
    float sv[2][ST::NI][2];
    int svrow[2];
    f32x4_t xr[4][NT];
    const int dlast = d0 + 16 * (nchunks - 1);
    constexpr int XA = 2;  // chunks X runs ahead
    stage.load(d0, sv[0], svrow[0]);
    xload(d0, xr[0]);
    if (XA == 2) xload(min(d0 + 16, dlast), xr[1]);
    if (nchunks > CPB) stage.load(d0 + 16 * CPB, sv[1], svrow[1]);
    stage.write(sb, sv[0], svrow[0]);
    barrier();
    u32x4_t avn[NM];  // first-product operands of the next chunk, see trip()
#pragma unroll
    for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(sb + fofs[j]);
    // LAG: what the second lane tile of the PREVIOUS chunk left -- its W*H, the second product's operand block, whether loop steps
    // beyond the range have to be masked.  In front of chunk 0 a dummy: ratios 0 * rcp(1) = 0, which add nothing to the numerators
    f32x4_t plag = {1.f, 1.f, 1.f, 1.f}, bnlag[NSA];
#pragma unroll
    for (int sn = 0; sn < NSA; ++sn) bnlag[sn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bool masklag = false, lagv = false;  // lagv: a chunk is behind (SSE: the dummy in front of chunk 0 has no residuals)
    int dchlag = d0;
    if (LAG) {
#pragma unroll
      for (int t = 0; t < NT; ++t) xr[3][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // one trip; FULLT: every chunk of the trip exists, every block of it has a successor after next and no chunk touches
    // the end of the loop range -> no guards in the unrolled body
    auto trip = [&](int c0, auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULLT = decltype(full_tag)::value;
      // the first product's operands of the NEXT chunk of a block are read from LDS right behind this chunk's first
      // product (into the registers it has just freed): their latency hides behind the ratios and the second product
      // instead of sitting in front of the next chunk's MFMAs.  (Not across a block's end: the other buffer is only
      // valid behind the barrier.)
#pragma unroll
      for (int ci = 0; ci < TRIP; ++ci) {
        const int c = c0 + ci;
        if (!FULLT && c >= nchunks) break;
        const int dch = d0 + 16 * c;
        const int buf = (ci / CPB) & 1, ch = ci % CPB;  // block parity (trips hold an even number of blocks)
        const bool more = FULLT || (c - ch + CPB < nchunks);       // a block follows the one this chunk belongs to
        const bool more2 = FULLT || (c - ch + 2 * CPB < nchunks);  // and one after that
        xload(FULLT ? dch + 16 * XA : min(dch + 16 * XA, dlast), xr[(ci + XA) & 3]);
        // (vmcnt retires in order: the staging requests go out AFTER this chunk's X prefetch)
        if (ch == 0 && more2) stage.load(dch + 32 * CPB, sv[buf], svrow[buf]);
        // the next block (requested a block ago) goes to the free LDS buffer BEFORE this block's last chunk: conversion
        // and LDS writes overlap with that chunk's matrix work instead of sitting in front of the barrier
        const bool last_of_block = ch == CPB - 1 || (!FULLT && c == nchunks - 1);
        if (last_of_block && more) stage.write(sb + (buf ^ 1) * ST::STB, sv[buf ^ 1], svrow[buf ^ 1]);
        __builtin_amdgcn_sched_barrier(0);  // loads stay in front of the arithmetic they overlap with
        // The block's barrier sits HERE, in front of its last chunk's arithmetic: every wave has written its part of
        // the next block and has fetched its last operands of this one (the second product's block below; the first
        // product's came with the previous chunk), so this buffer is free for the block after next and the last chunk
        // can already fetch the next block's first operands behind its first product.
        const char *b = sb + buf * ST::STB;
        f32x4_t bn[NSA];
#pragma unroll
        for (int sn = 0; sn < NSA; ++sn) {
          bn[sn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if (!OBJ) bn[sn] = *(const f32x4_t *)(b + nofs + sn * 64 + ch * ST::CHT);
        }
        if (last_of_block) barrier();
        if (!OBJ) {
          // The chunk's work skewed by lane tile (no register more): P(t0) | P(t1) + ratios(t0) | second product(t0) +
          // ratios(t1) | second product(t1).  The reciprocals of a tile wait only for that tile's first product, so they
          // issue in the free vector slots of the other tile's matrix instructions instead of behind all of them.
          u32x4_t av[NM];
#pragma unroll
          for (int j = 0; j < NM; ++j) av[j] = avn[j];  // fetched behind the previous chunk's first product
          f32x4_t p[NT], q[NT];
          const bool mask = !FULLT && dch + 16 > d1;
#define HYB_MFMA_THEN_RCP(n)                          \
  __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); \
  __builtin_amdgcn_sched_group_barrier(0x400, n, 0);
          if constexpr (LAG) {
            // P(t0) + reciprocals(t1 of the previous chunk) | P(t1) + reciprocals(t0) | multiplies of both | second products of both
            p_tile(0, av, p);
            f32x4_t ql;
            if (SSE) {  // the lagged tile's residuals (the objective this launch also leaves: see q_tile), ahead of its reciprocals
              float sqs = 0.0f;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float e = xr[(ci + 3) & 3][1][r] - plag[r];
                if (!FULLT && masklag) e = dchlag + 4 * g + r < d1 ? e : 0.0f;
                sqs = __builtin_fmaf(e, e, sqs);
              }
              spart += (lv[1] && lagv) ? sqs : 0.0f;
              asm volatile("" : "+v"(spart));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) ql[r] = __builtin_amdgcn_rcpf(plag[r]);
            if constexpr (SSE) {
            } else if constexpr (NM == 3) {
              HYB_MFMA_THEN_RCP(1) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(1)
            } else if constexpr (NM == 2) {
              HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(2)
            }
            __builtin_amdgcn_sched_barrier(0);
            p_tile(1, av, p);
            r_tile(0, dch, xr[ci & 3], p, q, mask);
            if constexpr (SSE) {
            } else if constexpr (NM == 3) {
              HYB_MFMA_THEN_RCP(1) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(1)
            } else if constexpr (NM == 2) {
              HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(2)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FULLT || c + 1 < nchunks) {
              const char *bnx = ch + 1 < CPB ? b + (ch + 1) * ST::CHP : sb + (buf ^ 1) * ST::STB;
#pragma unroll
              for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(bnx + fofs[j]);
            }
#pragma unroll
            for (int r = 0; r < 4; r += 2) {  // (the previous chunk's X entries: their register set is free until the next chunk's prefetch)
              const f32x2_t q2 = (f32x2_t){xr[(ci + 3) & 3][1][r], xr[(ci + 3) & 3][1][r + 1]} * (f32x2_t){ql[r], ql[r + 1]};
              ql[r] = q2.x;
              ql[r + 1] = q2.y;
              if (!FULLT && masklag) {
                ql[r] = (dchlag + 4 * g + r < d1) ? ql[r] : 0.0f;
                ql[r + 1] = (dchlag + 4 * g + r + 1 < d1) ? ql[r + 1] : 0.0f;
              }
            }
            m_tile(0, dch, xr[ci & 3], q, mask);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NS > 0) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int sn = 0; sn < NSA; ++sn) {
                  accs[1][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bnlag[sn][r], ql[r], accs[1][sn], 0, 0, 0);
                  accs[0][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bn[sn][r], q[0][r], accs[0][sn], 0, 0, 0);
                }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                accs[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bnlag[0][r], ql[r], accs[1][0], 0, 0, 0);
                accs[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0][r], q[0][r], accs[0][0], 0, 0, 0);
              }
            }
            plag = p[1];
#pragma unroll
            for (int sn = 0; sn < NSA; ++sn) bnlag[sn] = bn[sn];
            masklag = mask;
            dchlag = dch;
            lagv = true;
            if (SSE) sse_chunk();
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          p_tile(0, av, p);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 1; t < NT; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            p_tile(t, av, p);
            if constexpr (!LAGK && !SSE) {  // short loop ranges: rounds 3-4's order (ratios whole, the compiler's interleave) -- configs[1] 1.5 us faster with it
              q_tile(t - 1, dch, xr[ci & 3], p, q, mask);
              continue;
            }
            r_tile(t - 1, dch, xr[ci & 3], p, q, mask);
            if (!SSE) {  // the four reciprocals spread over the NM matrix instructions (0x8: MFMA, 0x400: transcendental)
              if constexpr (NM == 3) {
                HYB_MFMA_THEN_RCP(1) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(1)
              } else if constexpr (NM == 2) {
                HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(2)
              }
#undef HYB_MFMA_THEN_RCP
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (FULLT || c + 1 < nchunks) {  // (a following chunk at a block's end means a following block: `more`)
            const char *bnx = ch + 1 < CPB ? b + (ch + 1) * ST::CHP : sb + (buf ^ 1) * ST::STB;
#pragma unroll
            for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(bnx + fofs[j]);
          }
          if constexpr (LAGK || SSE) {
#pragma unroll
            for (int t = 0; t + 1 < NT; ++t) m_tile(t, dch, xr[ci & 3], q, mask);
          }
          q_tile(NT - 1, dch, xr[ci & 3], p, q, mask);
          f_tile(0, bn, q);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 1; t < NT; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            f_tile(t, bn, q);
          }
          if (SSE) sse_chunk();
          __builtin_amdgcn_sched_barrier(0);
        } else {
          u32x4_t av[NM];
#pragma unroll
          for (int j = 0; j < NM; ++j) av[j] = avn[j];  // fetched behind the previous chunk's first product
          f32x4_t p[NT];
          chunk_p(av, p);
          if (FULLT || c + 1 < nchunks) {  // (a following chunk at a block's end means a following block: `more`)
            const char *bnx = ch + 1 < CPB ? b + (ch + 1) * ST::CHP : sb + (buf ^ 1) * ST::STB;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(bnx + fofs[j]);
            __builtin_amdgcn_sched_barrier(0);
          }
          chunk_n(dch, xr[ci & 3], p, bn, !FULLT && dch + 16 > d1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    constexpr int AHEAD = 2 * CPB > 2 ? 2 * CPB : 2;
    int c0 = 0;
    for (; c0 + TRIP + AHEAD <= nchunks; c0 += TRIP) trip(c0, std::true_type());
    // the finish's old values: requested in front of the last trips (<= TRIP + AHEAD chunks) when they were not fetched before the loop
    if (fused && !AOLD_EARLY) {
      load_aold();
      aold_loaded = true;
    }
    for (; c0 < nchunks; c0 += TRIP) trip(c0, std::false_type());
    if constexpr (LAG) {  // the last chunk's second lane tile
      const int sl = (nchunks - 1) & 3;
      f32x4_t xl = sl == 0 ? xr[0][1] : sl == 1 ? xr[1][1] : sl == 2 ? xr[2][1] : xr[3][1];
      if (SSE) {
        float sqs = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = xl[r] - plag[r];
          if (masklag) e = dchlag + 4 * g + r < d1 ? e : 0.0f;
          sqs = __builtin_fmaf(e, e, sqs);
        }
        spart += lv[1] ? sqs : 0.0f;
        sse_chunk();
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = xl[r] * __builtin_amdgcn_rcpf(plag[r]);
        if (masklag) v = (dchlag + 4 * g + r < d1) ? v : 0.0f;
        if constexpr (NS > 0) {
#pragma unroll
          for (int sn = 0; sn < NSA; ++sn) accs[1][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bnlag[sn][r], v, accs[1][sn], 0, 0, 0);
        } else {
          accs[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bnlag[0][r], v, accs[1][0], 0, 0, 0);
        }
      }
    }
  };
  // objective: workgroup sum in wave order (fixed order => reproducible), one partial per workgroup
  auto sse_out = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_down(ssum, o, 64);
    double *sh = lds + 17 * 16;
    if (lane == 0) sh[wave] = ssum;
    __syncthreads();
    if (tid == 0) {
      double t = 0;
      for (int w = 0; w < nwaves; ++w) t += sh[w];
      ((double *)(arena + rdp->ossepart))[slot] = t * weight * weight;
    }
  };
  if (OBJ) {
    HybStage<KS, NMFK_HYB_CPB, 64 * NW, 0> stage;
    stage.init(B, k, D, tid);
    run(stage, sbase, [] { HYB_BARRIER(); });
    sse_out(tile);
    return;
  }
  if (ws == 1) {
    HybStage<KS, NMFK_HYB_CPB, 64 * NW, TM> stage;
    stage.init(B, k, D, tid);
    run(stage, sbase, [] { HYB_BARRIER(); });
  } else {
    HybStage<KS, 1, 64, TM> stage;
    stage.init(B, k, D, lane);
    run(stage, sbase + wave * 2 * HybStage<KS, 1, 64, TM>::STB, [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); });
    __syncthreads();  // the cross-wave scratch below overlays the staging buffers
  }
  if (SSE) sse_out(bx);
  // 4x4x1 form: a lane holds the sums over its own loop steps (d = 4g + r of every chunk) of ALL 4 NS signals: add the
  // four k-lane groups (lanes c16, c16 + 16, + 32, + 48: every lane gets the same bits), then lane group g keeps set g
  f32x4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (NS > 0) {
      acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sn = 0; sn < NSA; ++sn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = accs[t][sn][r];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (g == sn) acc[t][r] = v;
        }
    } else {
      acc[t] = accs[t][0];
    }
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
    // fuse_red: the W half-step divides by colsum(W) while its own workgroups already write the next one -- leave it a copy
    if (gp->fuse_red && which == 0 && bx == 0 && tid < k)
      ((float *)(arena + rdp->osnapW))[tid] = (float)nmfk_slot_sum<16>((const double *)(arena + rdp->osumW), k, rdp->nsW, tid);
    return;
  }

  if (!AOLD_EARLY && !aold_loaded) load_aold();
  __syncthreads();  // den[] is visible
  float *__restrict__ Anew = which == 0 ? (float *)(arena + NMFK_HOFF(*rdp, it + 1)) : (float *)(arena + rdp->oWt);
  double *sumA = (double *)(arena + (which == 0 ? rdp->osumH : rdp->osumW)) + (int64_t)tile * k;
  double *red = den + 16;  // [16][16]
  float vs[4] = {0.f, 0.f, 0.f, 0.f};
  bool low = false;  // a value below eps() written in a check iteration: the clamp (Mult:99-100) has work (NmfkState::lowflag)
  const float floorv = (gp->clampw > it + 1 && which == 1 && (it + 1) % 10 == 0) ? HYB_EPS : -__builtin_inff();
  if (owner) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (lv[t]) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * g + r;
          v[r] = 0.f;
          if (c < k) {
            v[r] = aold[t][r] * acc[t][r] / (float)den[c];  // Mult:67 / Mult:70 order
            v[r] = v[r] < floorv ? floorv : v[r];           // (NmfkStepArgs::clampw; a NaN stays)
            low = low || v[r] < HYB_EPS;
            if (!vec4) Anew[c + (int64_t)lt[t] * k] = v[r];
          }
          vs[r] += v[r];
        }
        if (vec4 && 4 * g < k) *(f32x4_t *)(Anew + 4 * g + (int64_t)lt[t] * k) = (f32x4_t){v[0], v[1], v[2], v[3]};
      }
  }
  if ((it + 1) % 10 == 0 && __any(low) && lane == 0) atomicOr(&const_cast<NmfkState *>(state)[u].lowflag, 1);
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
// RESIDENT form of the half-step (round 3): for a short loop dimension (the W half-step of a tall X: D = m = 512 at the
// BASELINE shape) the WHOLE loop factor, in both operand forms, fits in the LDS of a CU (D / 16 chunks of 2 - 2.8 KB).  The
// streaming kernel above pays, per workgroup of 8 waves and ONE pair of lane tiles per wave: the split of all D rows (at
// n = 8192 thirty-two workgroups of a unit each convert the same 512 rows), a barrier per 64 loop steps, the workgroup's
// prologue (denominators, first block) and an epilogue as long as the loop itself (PMC, profiles/r03: half of the W
// half-step's vector instructions sat outside the loop).  Here a workgroup of 16 waves stages the factor ONCE, and then
// every wave walks SEVERAL pairs of lane tiles on its own -- no barrier, no staging, operands straight from LDS, X two
// chunks ahead across tile boundaries -- with a slim per-tile epilogue (reciprocal denominators, sums kept per lane and
// reduced once per workgroup).  One sum-table slot per workgroup.  The loop runs in whole trips of four chunks: a D that is not a
// multiple of 64 (round 4: m = 1000 is as natural as 1024) is walked as roundup64(D) steps -- the rows behind D staged as zeros, their
// ratios (0 / 0) masked in the last trip.
// ------------------------------------------------------------------------------------------------------
#ifndef NMFK_HYB_RW
#define NMFK_HYB_RW 16  // waves per workgroup of the resident form
#endif
// OBJ: the monitored objective (Mult:74) of the units instead of a half-step, like the streaming kernel's OBJ mode: gp = the
// W half-step's arguments, `it` = parity of the H buffer that holds the current H, first product only, one partial per
// workgroup in ossepart[b] (the entries b >= gridDim.x up to ntile are zeroed: check_a_kernel adds ntile of them).
// SSE: the half-step also leaves the objective of the factors it reads (the streaming form's SSE mode; one partial per workgroup in
// ossepart[b], gridDim.x of them per unit)
template <int KS, int NS, int NT, bool OBJ, bool RAG, bool SSE>  // RAG: D is not a multiple of 64 (a kernel of its own: the masks cost the other 1-2 %)
__device__ __forceinline__ void hyb_res_body(char *arena, const float *__restrict__ Xt, const NmfkRun *__restrict__ runs,
                                             const NmfkState *__restrict__ state, const NmfkStepArgs *__restrict__ gp, int it, int u0,
                                             double weight, int ntile,
                                             double *lds) {  // lds: den[16], red[16*16], rden, then the factor: split planes [nch][CHP], transposed blocks [nch][CHT]
  constexpr int NM = KS == 16 ? 3 : KS == 8 ? 2 : 1;
  constexpr int NSA = NS > 0 ? NS : 1;
  constexpr int TM = OBJ ? 0 : (NS > 0 ? 2 : 1);
  constexpr int SUBMASK = KS == 16 ? 1 : 0;
  constexpr int RW = NMFK_HYB_RW;
  typedef HybStage<KS, 1, 64 * RW, TM> ST;  // (layout constants only)
  const int u = u0 + blockIdx.y, b = blockIdx.x, G = gridDim.x;
  if (!(gp->force && !OBJ) && !state[u].active) return;
  const NmfkRun *__restrict__ rdp = runs + u;
  const int k = rdp->k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int which = gp->which, L = gp->L, D = gp->D;
  const int Dp = (D + 63) & ~63;  // loop steps walked (whole trips)
  const int nch = Dp >> 4;
  constexpr bool ragged = RAG;    // the last trip holds steps >= D
  const float *__restrict__ A = (const float *)(arena + (which == 0 ? NMFK_HOFF(*rdp, it) : rdp->oWt));          // lane factor
  const float *__restrict__ B = (const float *)(arena + (which == 0 ? rdp->oWt : NMFK_HOFF(*rdp, OBJ ? it : it + 1)));  // loop factor
  char *sb = (char *)(lds + 18 * 16);  // den[16], red[16][16], rden[16] (+ padding)
  const int BFB = nch * ST::CHP;

  // ---- the whole loop factor -> LDS, once per workgroup: item = (row, pair of adjacent signals)
  // FR > 0 (W half-step behind an H half-step whose loop range was split FR ways over workgroups, NmfkStepArgs::fuse_red): the loop
  // factor H_new does not exist yet -- its rows are formed HERE from the partial numerators, in reduce_kernel's arithmetic and order
  // (sum of the partials in split order, H .* num ./ colsum(W): Mult:67), by every workgroup of the unit for itself; workgroup 0
  // also writes them, and rowsum(H) (below).  The reduce launch between the half-steps goes away.
  const int FR = (!OBJ && which == 1) ? gp->fuse_red : 0;
  double *den = lds;
  float *rden = (float *)(den + 16) + 16 * 32;  // behind red[16][16]; while the factor is staged: colsum(W) as fp32 (FR)
  if (FR) {
    if (tid < k) rden[tid] = ((const float *)(arena + rdp->osnapW))[tid];  // (osumW itself: written by the unit's workgroups that are done)
    __syncthreads();
  }
  double cs0 = 0.0, cs1 = 0.0;  // FR: sums of this thread's values of H_new (its items all have the same signal pair)
  for (int q = tid; q < Dp * ST::PPR; q += 64 * RW) {
    const int r = q / ST::PPR, cp = q - r * ST::PPR;
    float v0, v1;
    if (FR) {
      const int c0 = 2 * cp, c1 = c0 + 1;
      const bool in0 = r < D && c0 < k, in1 = r < D && c1 < k;
      const float *__restrict__ Hold = (const float *)(arena + NMFK_HOFF(*rdp, it));
      const float *__restrict__ part = (const float *)(arena + rdp->opart);
      float n0 = 0.0f, n1 = 0.0f;
      for (int s0 = 0; s0 < FR; s0 += 8) {  // eight partials in flight, added in the order of the splits
        float p0[8], p1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t e = ((int64_t)min(s0 + j, FR - 1) * D + min(r, D - 1)) * k;
          p0[j] = part[e + min(c0, k - 1)];
          p1[j] = part[e + min(c1, k - 1)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (s0 + j < FR) {
            n0 += p0[j];
            n1 += p1[j];
          }
      }
      const int64_t eo = (int64_t)min(r, D - 1) * k;
      v0 = in0 ? Hold[eo + c0] * n0 / rden[c0] : 0.0f;
      v1 = in1 ? Hold[eo + min(c1, k - 1)] * n1 / rden[min(c1, k - 1)] : 0.0f;
      if (b == 0) {
        float *Hn = const_cast<float *>(B);
        if (in0) Hn[eo + c0] = v0;
        if (in1) Hn[eo + c1] = v1;
      }
      cs0 += (double)v0;
      cs1 += (double)v1;
    } else {
      v0 = (r < D && 2 * cp < k) ? B[(int64_t)r * k + 2 * cp] : 0.0f;
      v1 = (r < D && 2 * cp + 1 < k) ? B[(int64_t)r * k + 2 * cp + 1] : 0.0f;
    }
    uint32_t h, m, l;
    split3_pair(v0, v1, h, m, l);
    const int ch = r >> 4, rr = r & 15;
    if (KS == 4) {
      char *d = sb + ch * ST::CHP + rr * 16 + cp * 4;
      *(uint32_t *)(d) = h;  // (h|h)
      *(uint32_t *)(d + 8) = h;
      *(uint32_t *)(d + 256) = m;  // (m|m)
      *(uint32_t *)(d + 256 + 8) = m;
      *(uint32_t *)(d + 512) = h;  // (h|l)
      *(uint32_t *)(d + 512 + 8) = l;
    } else {
      char *d = sb + ch * ST::CHP + (cp >> 2) * 256 + rr * 16 + (cp & 3) * 4;
      *(uint32_t *)(d) = h;
      *(uint32_t *)(d + ST::NH * 256) = m;
      *(uint32_t *)(d + 2 * ST::NH * 256) = l;
    }
    if (!OBJ) {
      char *t = sb + BFB + ch * ST::CHT + (rr >> 2) * ST::TPS + 2 * cp * 16 + (rr & 3) * 4;
      *(float *)(t) = v0;
      *(float *)(t + 16) = v1;
    }
  }
  double ssum = 0.0;
  if (FR) {
    // rowsum(H_new): lanes of equal signal pair within a wave (butterflies over the lane bits above the pair index), the waves in
    // order; every workgroup of the unit forms the same bits, workgroup 0 publishes them as slot 0 of the unit's table
    for (int o = 32; o >= ST::PPR; o >>= 1) {
      cs0 += __shfl_xor(cs0, o, 64);
      cs1 += __shfl_xor(cs1, o, 64);
    }
    double *redp = den + 16;  // [RW][16]
    if (lane < ST::PPR) {
      redp[wave * 16 + 2 * lane] = cs0;
      redp[wave * 16 + 2 * lane + 1] = cs1;
    }
    __syncthreads();
    if (tid < k) {
      double t = 0.0;
      for (int w = 0; w < RW; ++w) t += redp[w * 16 + tid];
      den[tid] = t;
      if (b == 0) {
        double *sumH = (double *)(arena + rdp->osumH);
        sumH[tid] = t;
        for (int pp = 1; pp < rdp->nsH; ++pp) sumH[pp * k + tid] = 0.0;
      }
    }
  } else if (!OBJ) {
    const double *sumB = (const double *)(arena + (which == 0 ? rdp->osumW : rdp->osumH));
    const int PB = which == 0 ? rdp->nsW : rdp->nsH;  // (the slots behind the unit's own are zero)
    if (tid < k) den[tid] = nmfk_slot_sum<4>(sumB, k, PB, tid);
  }
  __syncthreads();
  // 1 / sum(B) as fp32, once per workgroup (den[] is re-used for it: the finish of every tile pair reads it from LDS
  // instead of keeping four registers per lane alive across the loop)
  if (!OBJ) {
    if (tid < 16) rden[tid] = tid < k ? 1.0f / (float)den[tid] : 0.0f;
    __syncthreads();
  }

  const int ntp = (L + 16 * NT - 1) / (16 * NT);  // pairs (NT-tuples) of 16-lane tiles
  const int stride = RW * G;
  const int nD16 = (D + 15) >> 4, nL16 = (L + 15) >> 4;  // (chunks of the tiled copy of X: the chunks behind it read its last one again, masked)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)Xt, 0, -1, 0x00020000);
  int fofs[NM];
#pragma unroll
  for (int j = 0; j < NM; ++j) fofs[j] = (hyb_sa<KS>(j, g) * ST::NH + (g & SUBMASK)) * 256 + c16 * 16;
  const int nofs = BFB + g * ST::TPS + (NS > 0 ? (c16 & 3) : min(c16, KS - 1)) * 16;
  float *__restrict__ Anew = which == 0 ? (float *)(arena + NMFK_HOFF(*rdp, it + 1)) : (float *)(arena + rdp->oWt);
  // sums of the new factor: fp32 per lane over this wave's tile pairs (<= a few dozen values), fp64 from there on
  float vsf[4] = {0.f, 0.f, 0.f, 0.f};
  bool low = false;  // a value below eps() written in a check iteration (NmfkState::lowflag)
  const float floorv = (!OBJ && gp->clampw > it + 1 && which == 1 && (it + 1) % 10 == 0) ? HYB_EPS : -__builtin_inff();

  // X: the 16 x 16 block (16-lane tile, chunk) is 1 KB in lane order; byte offset = wave-uniform block offset (SGPR) + 16 * lane
  const uint32_t xlane = (uint32_t)lane * 16u;
  auto xoffs = [&](int tp, int (&xo)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) xo[t] = __builtin_amdgcn_readfirstlane(min(tp * NT + t, nL16 - 1) * nD16 * 1024);
  };
  auto xload = [&](const int (&xo)[NT], int c, f32x4_t (&xv)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      xv[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsx, xlane, xo[t] + (RAG ? min(c, nD16 - 1) : c) * 1024, 0));
  };
  f32x4_t xr[4][NT];
  u32x4_t avn[NM];
  int tp = b * RW + wave;
  int xo[NT], xn[NT];
  // Measured without gain (profiles/r03/resident_ablation.txt): the next pair's lane-factor rows and the current pair's old
  // values requested a tile, or four chunks, ahead of the boundary (19 spilled registers at 128 per wave; with 12 waves
  // per workgroup and 170 registers the kernel was slower as a whole), X three chunks ahead instead of two.
  HybRows nrow[NT];
  if (tp < ntp) {
    xoffs(tp, xo);
    xload(xo, 0, xr[0]);
    xload(xo, 1, xr[1]);
    if (NMFK_HYB_ABL & 1) {
      xload(xo, 2, xr[2]);
      xload(xo, 3, xr[3]);
    }
#pragma unroll
    for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(sb + fofs[j]);
  }
  for (; tp < ntp; tp += stride) {
    const int tnext = tp + stride < ntp ? tp + stride : tp;  // (the last pair prefetches itself again: dropped)
    xoffs(tnext, xn);
    bf16x8_t bop[NT][NM];
    int lt[NT];
    bool lv[NT];
    f32x4_t aoldv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int l = (tp * NT + t) * 16 + c16;
      lv[t] = l < L;
      lt[t] = lv[t] ? l : L - 1;
      nrow[t] = hyb_lane_rows<KS>(A, k, lt[t], g);
    }
    if (k == KS && (tp * NT + NT) * 16 <= L) {  // (wave-uniform) whole tiles of full-width rows: nothing to mask
#pragma unroll
      for (int t = 0; t < NT; ++t) hyb_lane_blocks_from<KS, NM>(nrow[t], k, true, g, bop[t], true);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) hyb_lane_blocks_from<KS, NM>(nrow[t], k, lv[t], g, bop[t]);
    }
    // (aold: one dword-aligned 16-byte load: a lane whose four signals end beyond the row reads into the next row -- the
    //  last row of the factor reads <= 12 bytes past it, inside the arena; those values are never used)
    auto load_aold = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < NT; ++t) aoldv[t] = *(const f32x4_u *)(A + min(4 * g, max(k - 1, 0)) + (int64_t)lt[t] * k);
    };
    f32x4_t accs[NT][NSA];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sn = 0; sn < NSA; ++sn) accs[t][sn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // four chunks per trip (X register sets and LDS offsets are compile-time); TAIL: the tile's last four chunks fetch
    // the first X entries of the wave's NEXT tile pair and the first operands of chunk 0 again
    auto trip = [&](int c0, auto tail_tag) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const int c = c0 + ci;
        if (TAIL && ci >= 2)
          xload(xn, ci - 2, xr[(ci + 2) & 3]);
        else
          xload(xo, c + 2, xr[(ci + 2) & 3]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t bn[NSA];
        if (!OBJ) {
#pragma unroll
          for (int sn = 0; sn < NSA; ++sn) bn[sn] = *(const f32x4_t *)(sb + nofs + sn * 64 + c * ST::CHT);
        }
        f32x4_t p[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, avn[j]), bop[t][j], p[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        {
          const char *nx = sb + ((TAIL && ci == 3) ? 0 : (c + 1) * ST::CHP);
#pragma unroll
          for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(nx + fofs[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (OBJ) {  // residuals: squares of a chunk in fp32 (packed), the chunk's partial into the fp64 sum (see the streaming form)
          float part = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            f32x2_t s2 = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              f32x2_t e2 = (f32x2_t){xr[ci][t][r], xr[ci][t][r + 1]} - (f32x2_t){p[t][r], p[t][r + 1]};
              if (TAIL && ragged) {
                e2.x = 16 * c + 4 * g + r < D ? e2.x : 0.0f;
                e2.y = 16 * c + 4 * g + r + 1 < D ? e2.y : 0.0f;
              }
              s2 = __builtin_elementwise_fma(e2, e2, s2);
            }
            part += lv[t] ? s2.x + s2.y : 0.0f;
          }
          ssum += (double)part;
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        f32x4_t q[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2_t rc = {__builtin_amdgcn_rcpf(p[t][r]), __builtin_amdgcn_rcpf(p[t][r + 1])};
            const f32x2_t q2 = (f32x2_t){xr[ci][t][r], xr[ci][t][r + 1]} * rc;
            q[t][r] = q2.x;
            q[t][r + 1] = q2.y;
            if (TAIL && ragged) {
              q[t][r] = 16 * c + 4 * g + r < D ? q[t][r] : 0.0f;
              q[t][r + 1] = 16 * c + 4 * g + r + 1 < D ? q[t][r + 1] : 0.0f;
            }
          }
        if (NS > 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int sn = 0; sn < NSA; ++sn)
#pragma unroll
              for (int t = 0; t < NT; ++t) accs[t][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bn[sn][r], q[t][r], accs[t][sn], 0, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) accs[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0][r], q[t][r], accs[t][0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // Software pipeline over the chunks (round 3, late; the same lag of one chunk in the streaming form's full trips needs
    // 12 more registers there: 22 spilled at 128 per wave, H half-step 0.682 -> 0.695 ms -- not kept): the first product of chunk c + 1 is issued BEFORE the ratios of
    // chunk c are formed, so that the reciprocals and multiplies of chunk c (which wait for nothing but chunk c's product,
    // finished a trip earlier) issue in the free vector slots of those matrix instructions instead of behind them.
    // pc: W*H of the chunk whose ratios are due; avn: the operands of the chunk after it.
    f32x4_t pc[NT];
    if (!OBJ) {
#pragma unroll
      for (int t = 0; t < NT; ++t) pc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          pc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, avn[j]), bop[t][j], pc[t], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(sb + ST::CHP + fofs[j]);
    }
    auto trip_pipe = [&](int c0, auto tail_tag) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const int c = c0 + ci;
        constexpr bool dummy = false;
        (void)dummy;
        const bool last = TAIL && ci == 3;  // (compile-time after unrolling) the tile pair's last chunk: nothing behind it
        if (!(NMFK_HYB_ABL & 1)) {
          if (TAIL && ci >= 2)
            xload(xn, ci - 2, xr[(ci + 2) & 3]);
          else
            xload(xo, c + 2, xr[(ci + 2) & 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t bn[NSA];
#pragma unroll
        for (int sn = 0; sn < NSA; ++sn) bn[sn] = (NMFK_HYB_ABL & 2) ? pc[0] : *(const f32x4_t *)(sb + nofs + sn * 64 + c * ST::CHT);
        f32x4_t pn[NT];
        if (!last) {
#pragma unroll
          for (int t = 0; t < NT; ++t) pn[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              pn[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, avn[j]), bop[t][j], pn[t], 0, 0, 0);
        }
        if (SSE) {  // residuals of chunk c (fp32 squares, the chunk's partial into the fp64 sum; anchored: see hyb_step_body)
          float spart = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float sqs = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float e = xr[ci][t][r] - pc[t][r];
              if (TAIL && ragged) e = 16 * c + 4 * g + r < D ? e : 0.0f;
              sqs = __builtin_fmaf(e, e, sqs);
            }
            spart += lv[t] ? sqs : 0.0f;
          }
          asm volatile("" : "+v"(spart));
          ssum += (double)spart;
        }
#ifndef NMFK_RES_OLD_ORDER
        // Where the ratios' instructions go (round 5, tools/probe/coissue2.hip, profiles/r05/issue_rates.txt; four waves per SIMD, one
        // stream): beside a v_mfma_f32_16x16x32_bf16 one v_rcp_f32 costs 2.3 cycles and two plain fp32 instructions nothing, but a
        // PACKED one (v_pk_mul_f32, v_pk_add_f32) 12; beside the fp32 matrix instructions everything adds, a packed multiply 5.5 for
        // its two values.  So: the reciprocals with the first product of chunk c + 1, the packed multiplies behind the barrier with
        // the second product of chunk c.
        f32x4_t q[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) q[t][r] = __builtin_amdgcn_rcpf(pc[t][r]);
        if (!last && !SSE) {  // the 4 NT reciprocals spread evenly over the NM NT matrix instructions (0x8: MFMA, 0x400: transcendental)
#define HYB_MFMA_THEN_RCP(n)                          \
  __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); \
  __builtin_amdgcn_sched_group_barrier(0x400, n, 0);
          if constexpr (NM * NT == 6) {
            HYB_MFMA_THEN_RCP(1) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(1) HYB_MFMA_THEN_RCP(1) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(1)
          } else if constexpr (NM * NT == 4) {
            HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(2) HYB_MFMA_THEN_RCP(2)
          } else if constexpr (NM * NT == 2) {
            HYB_MFMA_THEN_RCP(4) HYB_MFMA_THEN_RCP(4)
          }
#undef HYB_MFMA_THEN_RCP
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!last && !(NMFK_HYB_ABL & 2)) {  // operands of the chunk after next (the last-but-one chunk fetches chunk 0 again: the next tile pair's)
          const char *nx = sb + ((TAIL && ci == 2) ? 0 : (c + 2) * ST::CHP);
#pragma unroll
          for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(nx + fofs[j]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2_t q2 = (f32x2_t){xr[ci][t][r], xr[ci][t][r + 1]} * (f32x2_t){q[t][r], q[t][r + 1]};
            q[t][r] = q2.x;
            q[t][r + 1] = q2.y;
            if (TAIL && ragged) {  // (steps >= D: zero rows of the loop factor, 0 / 0)
              q[t][r] = 16 * c + 4 * g + r < D ? q[t][r] : 0.0f;
              q[t][r + 1] = 16 * c + 4 * g + r + 1 < D ? q[t][r + 1] : 0.0f;
            }
          }
#else  // (rounds 3-4's order: A/B builds only)
        f32x4_t q[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2_t rc = {__builtin_amdgcn_rcpf(pc[t][r]), __builtin_amdgcn_rcpf(pc[t][r + 1])};
            const f32x2_t q2 = (f32x2_t){xr[ci][t][r], xr[ci][t][r + 1]} * rc;
            q[t][r] = q2.x;
            q[t][r + 1] = q2.y;
            if (TAIL && ragged) {  // (steps >= D: zero rows of the loop factor, 0 / 0)
              q[t][r] = 16 * c + 4 * g + r < D ? q[t][r] : 0.0f;
              q[t][r + 1] = 16 * c + 4 * g + r + 1 < D ? q[t][r + 1] : 0.0f;
            }
          }
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {  // operands of the chunk after next (the last-but-one chunk fetches chunk 0 again: the next tile pair's)
          const char *nx = sb + ((TAIL && ci == 2) ? 0 : (c + 2) * ST::CHP);
#pragma unroll
          for (int j = 0; j < NM; ++j) avn[j] = *(const u32x4_t *)(nx + fofs[j]);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);  // (all multiplies ahead of the matrix instructions: split up, the allocator's register reuse costs wait states)
        if (NS > 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int sn = 0; sn < NSA; ++sn)
#pragma unroll
              for (int t = 0; t < NT; ++t) accs[t][sn] = __builtin_amdgcn_mfma_f32_4x4x1f32(bn[sn][r], q[t][r], accs[t][sn], 0, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) accs[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0][r], q[t][r], accs[t][0], 0, 0, 0);
        }
        if (!last) {
#pragma unroll
          for (int t = 0; t < NT; ++t) pc[t] = pn[t];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (!OBJ) {
      for (int c0 = 0; c0 + 4 < nch; c0 += 4) trip_pipe(c0, std::false_type());
      load_aold();  // the finish's old values: requested a trip ahead of their use
      trip_pipe(nch - 4, std::true_type());
    } else {
      for (int c0 = 0; c0 + 4 < nch; c0 += 4) trip(c0, std::false_type());
      trip(nch - 4, std::true_type());
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) xo[t] = xn[t];

    if (OBJ) continue;
    // ---- the tile pair's finish: A_new = A .* numerator ./ sum(B) (Mult:67 / Mult:70), sums of A_new per lane
    float vsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t acc;
      if (NS > 0) {
        acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sn = 0; sn < NSA; ++sn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = accs[t][sn][r];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == sn) acc[r] = v;
          }
      } else {
        acc = accs[t][0];
      }
      if (lv[t] && 4 * g < k) {
        const f32x4_t aold = aoldv[t];
        const f32x4_t rd4 = *(const f32x4_t *)(rden + 4 * g);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = 4 * g + r < k ? (aold[r] * acc[r]) * rd4[r] : 0.0f;
          v[r] = (4 * g + r < k && v[r] < floorv) ? floorv : v[r];  // (NmfkStepArgs::clampw; a NaN stays)
          low = low || (4 * g + r < k && v[r] < HYB_EPS);
          vsum[r] += v[r];
        }
        float *dst = Anew + 4 * g + (int64_t)lt[t] * k;
        if (4 * g + 3 < k) {
          *(f32x4_u *)dst = (f32x4_t){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if (4 * g + r < k) dst[r] = v[r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) vsf[r] += vsum[r];
  }

  if (OBJ) {  // workgroup sum in wave order (fixed order => reproducible), one partial per workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_down(ssum, o, 64);
    if (lane == 0) lds[wave] = ssum;
    __syncthreads();
    double *part = (double *)(arena + rdp->ossepart);
    if (tid == 0) {
      double t = 0;
      for (int w = 0; w < RW; ++w) t += lds[w];
      part[b] = t * weight * weight;
    }
    if (b == 0)
      for (int t = G + tid; t < ntile; t += 64 * RW) part[t] = 0.0;
    return;
  }
  if ((it + 1) % 10 == 0 && __any(low) && lane == 0) atomicOr(&const_cast<NmfkState *>(state)[u].lowflag, 1);
  if (SSE) {  // the workgroup's partial of the objective, waves in order (den[0..15] is free: the finishes read rden)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_down(ssum, o, 64);
    __syncthreads();
    if (lane == 0) lds[wave] = ssum;
    __syncthreads();
    if (tid == 0) {
      double t = 0;
      for (int w = 0; w < RW; ++w) t += lds[w];
      ((double *)(arena + rdp->ossepart))[b] = t * weight * weight;
    }
  }
  // ---- sums of the new factor over the workgroup's lane elements -> slot b (fixed order: lanes, then waves)
  double *red = den + 16;  // [RW][16]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = 4 * g + r;
    double v = (double)vsf[r];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (c16 == 0 && c < k) red[wave * 16 + c] = v;
  }
  __syncthreads();
  if (tid < k) {
    double t = red[tid];
    for (int w = 1; w < RW; ++w) t += red[w * 16 + tid];
    ((double *)(arena + (which == 0 ? rdp->osumH : rdp->osumW)))[(int64_t)b * k + tid] = t;
  }
}

// ------------------------------------------------------------------------------------------------------
// Wide ranks (16 < k <= 64; round 3): the split-operand first product where the split amortises.  mfma_wide_kernel
// (nmfk_step_impl.h) runs both products in plain fp32: kp/4 + kp/4 MFMAs of 32 cycles per 16 x 16 tile and 16 loop steps
// (k = 64: 1024 matrix cycles).  Here W*H comes from the three-term bf16 splits like above -- the six term pairs need
// 6 * KS contraction slots, i.e. 3 * NB instructions of 16 cycles for KS = 16 * NB signals (k = 64: 12 * 16 = 192 cycles
// instead of 512) -- and the numerators stay exact fp32 (4 * NB MFMAs of 32 cycles): 704 cycles per tile at k = 64,
// 352 at k <= 32.  Contraction layout: slot group G = 4 j + g (MFMA j, k-lane group g) <-> term pair G / (2 NB), block of
// eight signals G % (2 NB); a lane therefore only ever needs the blocks sub = g (NB = 2) or g, g + 4 (NB = 4) of its
// lane-factor row, in the three terms: 3 * NB / 2 operand registers of 128 bits per lane tile.
// Streaming form (a workgroup of 8 waves = 256 lane elements shares staged blocks of 64 loop rows, one barrier per block),
// loop range optionally split over workgroups (S > 1: partial numerators, reduce_kernel finishes).  2 waves per SIMD.
// ------------------------------------------------------------------------------------------------------
// OBJ: the monitored objective (Mult:74) instead of a half-step -- gp = the W half-step's arguments, `it` = parity of the H
// buffer that holds the current H, whole loop range, first product only, one partial per workgroup (256 rows of X) in
// ossepart[tile] like sse_kernel.  MODE 2: the half-step, and the objective of the factors it reads as a by-product (one partial
// per workgroup in ossepart[blockIdx.x]; see hyb_step_body's SSE mode and the deferred check of nmfk_mu_sweep).
// Order of the first product's matrix instructions in the BN form's explicit pipeline: slots that use the same A operand (plane of the loop
// factor) are adjacent, so that it is read from LDS once.  ord(slot) = the instruction index j of first_product's loops; load_of(slot) = running
// number of the slot's A operand; first(load) = the first slot that uses it; NL = operands per chunk.
template <int NB>
struct WideFpOrder {
  static constexpr int NM = 3 * NB, NL = NB == 2 ? 3 : 6;
  static constexpr int ord(int s) {
    constexpr int o2[6] = {0, 1, 4, 2, 3, 5};                              // term pairs (hh, hm, hl | mh, mm | lh)
    constexpr int o3[9] = {0, 1, 4, 2, 3, 5, 6, 7, 8};                     // the same for block g, then the three of block 4 + (g & 1)
    constexpr int o4[12] = {0, 2, 8, 1, 3, 9, 4, 6, 5, 7, 10, 11};         // j = 2 pair + block: h of block 0 (pairs 0, 1, 4), h of block 1, m, m, l, l
    return NB == 2 ? o2[s < 6 ? s : 0] : NB == 3 ? o3[s < 9 ? s : 0] : o4[s < 12 ? s : 0];
  }
  static constexpr int first(int l) {
    constexpr int f2[3] = {0, 3, 5}, f3[6] = {0, 3, 5, 6, 7, 8}, f4[6] = {0, 3, 6, 8, 10, 11};
    return NB == 2 ? f2[l < 3 ? l : 0] : NB == 3 ? f3[l < 6 ? l : 0] : f4[l < 6 ? l : 0];
  }
  static constexpr int load_of(int s) {
    int l = 0;
    for (int i = 1; i < NL; ++i)
      if (first(i) <= s) l = i;
    return l;
  }
};
// BN (round 6): the numerators on the bf16 pipe (see hyb_pack_hi / hyb_trunc_rest above): the second product's operand block is staged as bf16 term
// planes per block of 16 signals, loop steps contiguous ([term][k-lane group g][signal][4 steps] = 8 bytes per lane).
template <int NB, int NT, int MODE, bool BN = false>
__global__ __launch_bounds__(512, 2) void wide2_step_kernel(char *arena, const float *__restrict__ Xt, const NmfkRun *__restrict__ runs,
                                                            const NmfkState *__restrict__ state, const NmfkStepArgs *__restrict__ gp,
                                                            int it, int u0, double weight) {
  extern __shared__ double lds[];  // den[64], red[8][64], then two staged blocks
  constexpr bool OBJ = MODE == 1, SSE = MODE == 2;
  constexpr int KS = 16 * NB, NM = 3 * NB, NH = 2 * NB, CPB = 4;
  static_assert(!(BN && MODE == 1), "the objective mode has no second product");
  typedef WideFpOrder<NB> FPL;
  constexpr int CHP = 3 * NH * 256, CHT = BN ? NB * HYB_BN_BLK : NB * 4 * 256;  // bytes of a chunk's split planes / transposed blocks
  constexpr int BFB = CPB * CHP, STB = BFB + CPB * CHT;
  constexpr int PPR = KS / 2, NITEM = 16 * CPB * PPR, NI = NITEM / 512;  // staging items (row, signal pair) per thread
  constexpr int NSUB = NB == 3 ? 1 : NB / 2;  // blocks of eight signals of the lane factor a lane needs per term (sub = g + 4 i)
  static_assert(NB == 2 || NB == 3 || NB == 4, "KS = 32, 48 or 64");
  // NB = 3 (KS = 48, round 4): six blocks of eight signals do not divide over the four k-lane groups, so the 36 (term pair,
  // block) groups are laid out as  MFMA j < 6: pair j, block g;  MFMA 6 + q (q < 3): pair 2q + (g >> 1), block 4 + (g & 1).
  // A lane then needs block g in the three terms and block 4 + (g & 1) in two (pairs 0, 2, 4 want the lane factor's terms
  // h, h, l -- k-lane groups 0, 1 --, pairs 1, 3, 5 want m, m, h -- groups 2, 3): five operand registers of 128 bits per tile.
  static_assert(NITEM % 512 == 0, "items divide evenly");
  // Which (lane tile / split, unit) this workgroup serves.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest).  Taken as they
  // come -- x = tile, y = unit -- the chip works through ONE unit at a time, and at BASELINE configs[4]'s size (X = 537 MB: neither the L2s nor
  // the 256 MB Infinity Cache hold it) every unit streams all of X from HBM (TCC hit 50 %).  Remapped so that, on every XCD, the UNITS of a tile run
  // side by side (unit fastest within the XCD's share of the tiles): the restarts walk the same tile of X chunk by chunk at the same pace and all but
  // the first find it in that XCD's L2 (k = 64: 3.10 -> 2.95 ms per iteration; the fp32-numerator form 3.65 -> 3.48).
  int bx = blockIdx.x, by = blockIdx.y;
  if ((gridDim.x & 7) == 0 && gridDim.y > 1) {
    const int lin = by * gridDim.x + bx, j = lin >> 3;
    by = j % (int)gridDim.y;
    bx = (j / (int)gridDim.y) * 8 + (lin & 7);
  }
  const int u = u0 + by;
  if (!(gp->force && !OBJ) && !state[u].active) return;
  const NmfkRun *__restrict__ rdp = runs + u;
  const int k = rdp->k, kp = rdp->kp;  // true rank, row stride (padding rows of the factors are zero and stay zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int which = gp->which, S = OBJ ? 1 : gp->S, L = gp->L, D = gp->D;
  const int tile = bx / S, sp = bx - tile * S;
  const int l0 = tile * (16 * NT * 8) + wave * 16 * NT;
  const float *__restrict__ A = (const float *)(arena + (which == 0 ? NMFK_HOFF(*rdp, it) : rdp->oWt));      // lane factor
  const float *__restrict__ B = (const float *)(arena + (which == 0 ? rdp->oWt : NMFK_HOFF(*rdp, OBJ ? it : it + 1)));  // loop factor
  const int d0 = __builtin_amdgcn_readfirstlane(OBJ ? 0 : sp * gp->dchunk);
  const int d1 = __builtin_amdgcn_readfirstlane(OBJ ? D : min(D, d0 + gp->dchunk));
  double ssum = 0.0;
  const int nchunks = (d1 - d0 + 15) >> 4;
  constexpr int tA[6] = {0, 0, 1, 1, 0, 2}, tB[6] = {0, 1, 0, 1, 2, 0};  // term pairs (hh, hm, mh, mm, hl, lh)

  // lane-factor operand blocks: bopt[t][term][i] = signals [8 (g + 4 i), + 8) of row l in term `term`
  bf16x8_t bopt[NT][3][NSUB];
  bf16x8_t bq01[NT], bq2[NT];  // (NB = 3) block 4 + (g & 1): the term of the MFMAs 6, 7 and of the MFMA 8
  int lt[NT];
  bool lv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int l = l0 + 16 * t + c16;
    lv[t] = l < L;
    lt[t] = lv[t] ? l : L - 1;
    if (NB == 3) {
      const int s0 = 8 * (4 + (g & 1));
      const float *rp = A + (int64_t)lt[t] * kp + (s0 < kp ? s0 : 0);
      const f32x4_t r0 = *(const f32x4_u *)rp, r1 = *(const f32x4_u *)(rp + 4);
      u32x4_t hh, mm, ll;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 2 * w + e;
          v[e] = (lv[t] && s0 + c < kp) ? (c < 4 ? r0[c & 3] : r1[c & 3]) : 0.0f;
        }
        uint32_t h, m, lo;
        split3_pair(v[0], v[1], h, m, lo);
        hh[w] = h;
        mm[w] = m;
        ll[w] = lo;
      }
      const bool up = g >= 2;
      u32x4_t w01, w2;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        w01[w] = up ? mm[w] : hh[w];
        w2[w] = up ? hh[w] : ll[w];
      }
      bq01[t] = __builtin_bit_cast(bf16x8_t, w01);
      bq2[t] = __builtin_bit_cast(bf16x8_t, w2);
    }
#pragma unroll
    for (int i = 0; i < NSUB; ++i) {
      const int s0 = 8 * (g + 4 * i);
      const float *rp = A + (int64_t)lt[t] * kp + (s0 < kp ? s0 : 0);  // (a block that ends beyond the row reads into the next row: masked)
      const f32x4_t r0 = *(const f32x4_u *)rp, r1 = *(const f32x4_u *)(rp + 4);
      u32x4_t hh, mm, ll;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 2 * w + e;
          v[e] = (lv[t] && s0 + c < kp) ? (c < 4 ? r0[c & 3] : r1[c & 3]) : 0.0f;
        }
        uint32_t h, m, lo;
        split3_pair(v[0], v[1], h, m, lo);
        hh[w] = h;
        mm[w] = m;
        ll[w] = lo;
      }
      bopt[t][0][i] = __builtin_bit_cast(bf16x8_t, hh);
      bopt[t][1][i] = __builtin_bit_cast(bf16x8_t, mm);
      bopt[t][2][i] = __builtin_bit_cast(bf16x8_t, ll);
    }
  }
  f32x4_t acc[NT][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // X from the tiled copy: block (16-lane tile, chunk) = 1 KB in lane order
  const int nD16 = (D + 15) >> 4, nL16 = (L + 15) >> 4;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)Xt, 0, -1, 0x00020000);
  const uint32_t xlane = (uint32_t)lane * 16u;
  int xo[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xo[t] = __builtin_amdgcn_readfirstlane(min((l0 >> 4) + t, nL16 - 1) * nD16 * 1024);
  auto xload = [&](int dch, f32x4_t (&xv)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      xv[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsx, xlane, xo[t] + dch * 64, 0));
  };

  // staging: item q = tid + 512 i -> (row r of the block, signal pair cp); the rows of a block are contiguous in memory
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void *)B, 0, -1, 0x00020000);
  char *sb = (char *)(lds + 9 * NMFK_MAX_K);
  float sv[NI][2];
  // Item i of a thread -> (row r of the block, signal pair cp).  Round 6: the 32 lanes of a half wave take 8 rows x 4 adjacent pairs of one plane
  // (J = the (row group, plane) the half wave serves), so that their ds_write_b32 of a split plane land on 128 contiguous bytes = 32 different banks;
  // with the lanes along a row (rounds 3-5: q = tid + 512 i, r = q / PPR) eight planes of 256 B met on four banks (8-way conflicts; half of the
  // LDS's busy cycles were bank conflicts, profiles/r06/wide_pmc_summary_bn1.txt).  The memory side reads 32-byte pieces of eight rows instead of whole
  // rows: the factor's rows are L2-resident and a block is NI load instructions per thread.
  auto stage_item = [&](int i, int &r, int &cp) __attribute__((always_inline)) {
    const int J = ((tid >> 6) * NI + i) * 2 + ((tid >> 5) & 1);  // 0 .. 2 PPR - 1
    const int rh = J / (PPR / 4), cph = J - rh * (PPR / 4);
    r = 8 * rh + ((tid >> 2) & 7);
    cp = 4 * cph + (tid & 3);
  };
  static_assert(NI * 8 == PPR, "eight waves x NI items x two half waves = 2 PPR (row group, plane) pairs");
  auto stage_load = [&](int row0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int r, cp;
      stage_item(i, r, cp);
      const uint32_t vo = (uint32_t)((r * kp + min(2 * cp, max(kp - 2, 0))) * 4);
      const f32x2_t v = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsb, vo, row0 * kp * 4, 0));
      sv[i][0] = v.x;
      sv[i][1] = v.y;
    }
  };
  auto stage_write_item = [&](char *dst, int row0, int i) __attribute__((always_inline)) {
    {
      int r, cp;
      stage_item(i, r, cp);
      const bool ok = row0 + r < D;  // (rows past the factor's end and padding signals are staged as zeros)
      const float v0 = (ok && 2 * cp < kp) ? sv[i][0] : 0.0f, v1 = (ok && 2 * cp + 1 < kp) ? sv[i][1] : 0.0f;
      uint32_t h, m, lo;
      split3_pair(v0, v1, h, m, lo);
      const int ch = r >> 4, rr = r & 15;
      char *d = dst + ch * CHP + (cp >> 2) * 256 + rr * 16 + (cp & 3) * 4;  // plane (term, block of eight signals cp >> 2)
      *(uint32_t *)(d) = h;
      *(uint32_t *)(d + NH * 256) = m;
      *(uint32_t *)(d + 2 * NH * 256) = lo;
      // transposed: plane (block of sixteen signals, loop steps [4g, 4g + 4)), signal c at (c & 15) * 16: four fp32 values
      if (!OBJ && BN) {  // planes (b_l | b_h) and (b_h | b_m) of a block of 16 signals: (k-lane group rr >> 2, signal c) at 16 (16 (rr >> 2) + c), step rr & 3 at + 2 (rr & 3)
        char *tr = dst + BFB + ch * CHT + ((2 * cp) >> 4) * HYB_BN_BLK + ((rr >> 2) * 16 + ((2 * cp) & 15)) * 16 + (rr & 3) * 2;
        *(uint16_t *)(tr) = (uint16_t)lo;
        *(uint16_t *)(tr + 16) = (uint16_t)(lo >> 16);
        *(uint16_t *)(tr + 8) = (uint16_t)h;
        *(uint16_t *)(tr + 8 + 16) = (uint16_t)(h >> 16);
        *(uint16_t *)(tr + 1024) = (uint16_t)h;
        *(uint16_t *)(tr + 1024 + 16) = (uint16_t)(h >> 16);
        *(uint16_t *)(tr + 1024 + 8) = (uint16_t)m;
        *(uint16_t *)(tr + 1024 + 8 + 16) = (uint16_t)(m >> 16);
      } else if (!OBJ) {
        char *tr = dst + BFB + ch * CHT + (((2 * cp) >> 4) * 4 + (rr >> 2)) * 256 + ((2 * cp) & 15) * 16 + (rr & 3) * 4;
        *(float *)(tr) = v0;
        *(float *)(tr + 16) = v1;
      }
    }
  };
  auto stage_write = [&](char *dst, int row0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) stage_write_item(dst, row0, i);
  };

  // inputs of the fused finish, requested before the loop: the other factor's sum table (denominators of Mult:67 / Mult:70).
  // Up to 4096 entries (slots x kp; 2048 for KS = 32) travel in eight (four) registers per thread and go through LDS behind the loop, where thread c adds
  // the slots of signal c in slot order; one thread loading its 64 slots one after the other there -- a chain of L2 round trips
  // with nothing else on the CU to hide it (one workgroup per CU) -- was a sixth of the W half-step at k = 64.
  const bool fusedf = !OBJ && gp->fused != 0;
  const double *sumB = (const double *)(arena + (which == 0 ? rdp->osumW : rdp->osumH));
  const int PB = which == 0 ? gp->PW : gp->PH;
  const int ntab = PB * kp;
  constexpr int NTAB = NB == 2 ? 4 : 8;  // (KS = 32: 64 slots x 32 signals; four more registers would cost that form its second workgroup per CU)
  const bool tabpre = fusedf && ntab <= 512 * NTAB;
  double tabv[NTAB];
  if (tabpre) {
#pragma unroll
    for (int i = 0; i < NTAB; ++i) tabv[i] = tid + 512 * i < ntab ? sumB[tid + 512 * i] : 0.0;
  }
  f32x4_t xr[4][NT];
  // X runs XA chunks ahead of its use in four register sets (a chunk's set is free once its ratios are formed).  Compiled out, the X loads are
  // 4-5 % of the half-step at BASELINE configs[4]'s size, 1 % of it the misses (profiles/r06/wide_ablation.txt); three chunks ahead instead of two
  // changes nothing at 48 / 64 signals and costs the 32-signal form its registers.
  constexpr int XA = NMFK_WIDE_XA;
  if (nchunks > 0) {
    const int dlast = d0 + 16 * (nchunks - 1);
    stage_load(d0);
    xload(d0, xr[0]);
    xload(min(d0 + 16, dlast), xr[1]);
    if (XA == 3) xload(min(d0 + 32, dlast), xr[2]);
    if (NMFK_WIDE_ABL & 2) {
      xload(d0, xr[2]);
      xload(d0, xr[3]);
    }
    stage_write(sb, d0);
    // first product: P[d = 4g + r][l = c16] from the six term pairs
    auto first_product = [&](const char *cur, int chx, f32x4_t (&p)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) p[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (NB == 3) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {  // pair j, block g
        const bf16x8_t av = *(const bf16x8_t *)(cur + chx * CHP + (tA[j] * NH + g) * 256 + c16 * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bopt[t][tB[j]][0], p[t], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {  // pair 2q + (g >> 1), block 4 + (g & 1): the loop factor's term is 0, 1, (g >> 1 ? 2 : 0)
        const int ta = q < 2 ? q : (g >= 2 ? 2 : 0);
        const bf16x8_t av = *(const bf16x8_t *)(cur + chx * CHP + (ta * NH + 4 + (g & 1)) * 256 + c16 * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, q < 2 ? bq01[t] : bq2[t], p[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        const int tp = NB == 4 ? (j >> 1) : j, i = NB == 4 ? (j & 1) : 0;  // group G = 4j + g: term pair, block sub = g + 4 i
        const bf16x8_t av = *(const bf16x8_t *)(cur + chx * CHP + (tA[tp] * NH + 4 * i + g) * 256 + c16 * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bopt[t][tB[tp]][i], p[t], 0, 0, 0);
      }
    }
    };
    // the same operands one at a time, for the explicit pipeline of the BN form: A operand of matrix instruction j of chunk chx, B operand of tile t
    auto fp_av = [&](const char *cur, int chx, int j) __attribute__((always_inline)) -> bf16x8_t {
      int plane;
      if (NB == 3) {
        const int q = j - 6;
        plane = j < 6 ? tA[j < 6 ? j : 0] * NH + g : ((q < 2 ? q : (g >= 2 ? 2 : 0)) * NH + 4 + (g & 1));
      } else {
        const int tp = NB == 4 ? (j >> 1) : j, i = NB == 4 ? (j & 1) : 0;
        plane = tA[tp] * NH + 4 * i + g;
      }
      if (NMFK_WIDE_ABL & 16) return bopt[0][0][0];
      return *(const bf16x8_t *)(cur + chx * CHP + plane * 256 + c16 * 16);
    };
    auto fp_b = [&](int t, int j) __attribute__((always_inline)) -> bf16x8_t {
      if (NB == 3) return j < 6 ? bopt[t][tB[j < 6 ? j : 0]][0] : (j < 8 ? bq01[t] : bq2[t]);
      const int tp = NB == 4 ? (j >> 1) : j, i = NB == 4 ? (j & 1) : 0;
      return bopt[t][tB[tp]][i];
    };
    // second product's operand forms of block nb of chunk chx: plane 0 = (b_l | b_h), plane 1 = (b_h | b_m)
    auto sp_av = [&](const char *cur, int chx, int nb, int plane) __attribute__((always_inline)) -> bf16x8_t {
      if (NMFK_WIDE_ABL & 16) return bopt[0][1][0];
      return *(const bf16x8_t *)(cur + BFB + chx * CHT + nb * HYB_BN_BLK + plane * 1024 + (g * 16 + c16) * 16);
    };
    const int nblocks = (nchunks + CPB - 1) / CPB;
    f32x4_t pnext[NT];  // PIPE: W*H of the block's next chunk (see the chunk loop)
#pragma unroll
    for (int t = 0; t < NT; ++t) pnext[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ringF[4];  // BN: A operands of the first product in flight (see the chunk loop)
    for (int blk = 0; blk < nblocks; ++blk) {
      char *cur = sb + (blk & 1) * STB, *nxt = sb + ((blk & 1) ^ 1) * STB;
      const bool more = blk + 1 < nblocks;
      if (more) stage_load(d0 + 64 * (blk + 1));
      __syncthreads();  // this block is staged (and the other buffer is free: everybody left it a block ago)
#pragma unroll
      for (int ch = 0; ch < CPB; ++ch) {
        const int c = blk * CPB + ch;
        if (c >= nchunks) break;
        const int dch = d0 + 16 * c;
        if (BN && (NMFK_WIDE_ABL & 32))  // (X loads issued as always, but from the same four chunks of the tile over and over: cache hits)
          xload(d0 + 16 * ((ch + XA) & 3), xr[(ch + XA) & 3]);
        else if (!(BN && (NMFK_WIDE_ABL & 2)))
          xload(min(dch + 16 * XA, dlast), xr[(ch + XA) & 3]);
        __builtin_amdgcn_sched_barrier(0);
        // PIPE (round 5; NB >= 3, half-step modes): inside a staged block the first product of chunk ch + 1 is issued BEFORE the ratios of
        // chunk ch, whose reciprocals then sit beside bf16 matrix instructions (2-6 cycles each instead of 8 beside the fp32 ones:
        // profiles/r05/issue_rates.txt).  Not across a block's end (the other LDS buffer is being rewritten), not at 32 signals (8 more
        // registers would take that instantiation from four waves per SIMD to three).
        constexpr bool PIPE = NB >= 3 && !OBJ;
        if constexpr (BN) {
          // ---- numerators on the bf16 pipe (round 6): an EXPLICIT software pipeline.  With two waves per SIMD nothing hides an LDS round trip
          // or a chain of vector instructions, so the chunk is laid out slot by slot (a slot = one A operand, NT matrix instructions; a fence behind
          // each): phase 1 = the first product of the block's NEXT chunk, with the reciprocals, multiplies and three-term split of THIS chunk's
          // ratios dealt over its slots (12 pieces: one per ratio, one per pair of operand words); phase 2 = this chunk's second product.  Every A
          // operand is requested from LDS three slots before its matrix instructions (a ring of four registers for the first product; the second
          // product's two forms double-buffered by block of 16 signals).  The first product's matrix instructions run in the order that keeps equal A
          // operands together (the loop factor's h term serves three term pairs, m two, l one: WideFpOrder), so an operand is read once.
          constexpr int NPIECE = 6 * NT;
          const bool next = ch + 1 < CPB, next2 = ch + 2 < CPB;  // (compile-time after unrolling; products of chunks behind the range's end are computed from staged rows and dropped)
          const bool edge = dch + 16 > d1;
          f32x4_t p[NT];
          if (ch == 0) {
            first_product(cur, 0, p);
            ringF[0] = fp_av(cur, 1, FPL::ord(0));
          } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) p[t] = pnext[t];
          }
          if (SSE) {  // the residuals of the chunk (fp32 squares, the chunk's partial into the fp64 sum: the OBJ mode's error budget)
            float part = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              float sqs = 0.0f;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float e = xr[ch & 3][t][r] - p[t][r];
                e = (!edge || dch + 4 * g + r < d1) ? e : 0.0f;
                sqs = __builtin_fmaf(e, e, sqs);
              }
              part += lv[t] ? sqs : 0.0f;
            }
            asm volatile("" : "+v"(part));  // (anchor: see hyb_step_body)
            ssum += (double)part;
          }
          float qv[NT][4], r1[NT][4], r2[NT][4];
          uint32_t hw[NT][2], mw[NT][2], lw[NT][2];
          if (NMFK_WIDE_ABL & 4) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int e = 0; e < 2; ++e) hw[t][e] = mw[t][e] = lw[t][e] = 0x3f803f80u + lane;
          }
          auto piece = [&](int i) __attribute__((always_inline)) {
            if (i < 4 * NT) {
              const int t = i >> 2, r = i & 3;
              float v = xr[ch & 3][t][r] * __builtin_amdgcn_rcpf(p[t][r]);
              v = (!edge || dch + 4 * g + r < d1) ? v : 0.0f;  // loop steps beyond the range give zero
              qv[t][r] = v;
              r1[t][r] = hyb_trunc_rest(v);
              r2[t][r] = hyb_trunc_rest(r1[t][r]);
            } else {
              const int e = i - 4 * NT, t = e >> 1, hf = e & 1;
              hw[t][hf] = hyb_pack_hi(qv[t][2 * hf], qv[t][2 * hf + 1]);
              mw[t][hf] = hyb_pack_hi(r1[t][2 * hf], r1[t][2 * hf + 1]);
              lw[t][hf] = hyb_pack_hi(r2[t][2 * hf], r2[t][2 * hf + 1]);
            }
          };
          bf16x8_t ahm[2], alh[2];
          if (next) {
#pragma unroll
            for (int sl = 0; sl < NM; ++sl) {
#pragma unroll
              for (int ld = 1; ld < FPL::NL; ++ld)
                if (FPL::first(ld) - 3 == sl || (sl == 0 && FPL::first(ld) < 3)) ringF[ld & 3] = fp_av(cur, ch + 1, FPL::ord(FPL::first(ld)));
              if (sl + 3 == NM) ahm[0] = sp_av(cur, ch, 0, 1);
              if (sl + 2 == NM) alh[0] = sp_av(cur, ch, 0, 0);
#pragma unroll
              for (int t = 0; t < NT; ++t)
                pnext[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ringF[FPL::load_of(sl) & 3], fp_b(t, FPL::ord(sl)), sl == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : pnext[t], 0, 0, 0);
#pragma unroll
              for (int i = sl * NPIECE / NM; i < (sl + 1) * NPIECE / NM; ++i)
                if (!(NMFK_WIDE_ABL & 4) || blk == 0) piece(i);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {  // the block's last chunk: nothing to put the ratios beside
            ahm[0] = sp_av(cur, ch, 0, 1);
            alh[0] = sp_av(cur, ch, 0, 0);
#pragma unroll
            for (int i = 0; i < NPIECE; ++i)
              if (!(NMFK_WIDE_ABL & 4) || blk == 0) piece(i);
            __builtin_amdgcn_sched_barrier(0);
          }
          HybQ qs[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            qs[t].hh = __builtin_bit_cast(bf16x8_t, (u32x4_t){hw[t][0], hw[t][1], hw[t][0], hw[t][1]});
            qs[t].mm = __builtin_bit_cast(bf16x8_t, (u32x4_t){mw[t][0], mw[t][1], mw[t][0], mw[t][1]});
            qs[t].hl = __builtin_bit_cast(bf16x8_t, (u32x4_t){hw[t][0], hw[t][1], lw[t][0], lw[t][1]});
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const bool lastb = nb + 1 == NB;
            // the NEXT block's rows (requested at this block's start) are converted and written to the other LDS buffer HERE, beside the second
            // product of the block's third chunk -- a phase with no vector work of its own -- instead of behind the block with the matrix pipe idle
            static_assert(NI <= NB, "one staging item per block of 16 signals");
            if (more && ch == CPB - 2 && nb < NI && !(NMFK_WIDE_ABL & 1)) stage_write_item(nxt, d0 + 64 * (blk + 1), nb);
            if (!lastb) ahm[(nb + 1) & 1] = sp_av(cur, ch, nb + 1, 1);
            else if (next2) ringF[0] = fp_av(cur, ch + 2, FPL::ord(0));
#pragma unroll
            for (int t = 0; t < NT; ++t)
              if (!(NMFK_WIDE_ABL & 8)) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[nb & 1], qs[t].hh, acc[t][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!lastb) alh[(nb + 1) & 1] = sp_av(cur, ch, nb + 1, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
              if (!(NMFK_WIDE_ABL & 8)) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[nb & 1], qs[t].mm, acc[t][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
              if (!(NMFK_WIDE_ABL & 8)) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alh[nb & 1], qs[t].hl, acc[t][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          continue;
        }
        f32x4_t p[NT];
        if (!PIPE || ch == 0) first_product(cur, ch, p);
        if (PIPE && ch == 0) __builtin_amdgcn_sched_barrier(0);  // (the group barriers below pair the NEXT chunk's matrix instructions with the reciprocals)
        if (PIPE && ch > 0) {
#pragma unroll
          for (int t = 0; t < NT; ++t) p[t] = pnext[t];
        }
        const bool edge = dch + 16 > d1;
        if (OBJ) {  // residuals: squares of a chunk in fp32 (packed), the chunk's partial into the fp64 sum (see hyb_step_body)
          float part = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            f32x2_t s2 = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              f32x2_t e2 = (f32x2_t){xr[ch & 3][t][r], xr[ch & 3][t][r + 1]} - (f32x2_t){p[t][r], p[t][r + 1]};
              e2.x = (!edge || dch + 4 * g + r < d1) ? e2.x : 0.0f;
              e2.y = (!edge || dch + 4 * g + r + 1 < d1) ? e2.y : 0.0f;
              s2 = __builtin_elementwise_fma(e2, e2, s2);
            }
            part += lv[t] ? s2.x + s2.y : 0.0f;
          }
          ssum += (double)part;
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        if (SSE) {  // the residuals of the chunk (fp32 squares, the chunk's partial into the fp64 sum: the OBJ mode's error budget)
          float part = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float sqs = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float e = xr[ch & 3][t][r] - p[t][r];
              e = (!edge || dch + 4 * g + r < d1) ? e : 0.0f;
              sqs = __builtin_fmaf(e, e, sqs);
            }
            part += lv[t] ? sqs : 0.0f;
          }
          asm volatile("" : "+v"(part));  // (anchor: see hyb_step_body)
          ssum += (double)part;
        }
        // ratios; loop steps beyond the range give zero.  PIPE: the reciprocals together with the first product of the block's next chunk
        f32x4_t q[NT];
        if constexpr (!PIPE) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const f32x2_t rc = {__builtin_amdgcn_rcpf(p[t][r]), __builtin_amdgcn_rcpf(p[t][r + 1])};
              const f32x2_t q2 = (f32x2_t){xr[ch & 3][t][r], xr[ch & 3][t][r + 1]} * rc;
              q[t][r] = (!edge || dch + 4 * g + r < d1) ? q2.x : 0.0f;
              q[t][r + 1] = (!edge || dch + 4 * g + r + 1 < d1) ? q2.y : 0.0f;
            }
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) q[t][r] = __builtin_amdgcn_rcpf(p[t][r]);
        // (no run-time condition here: reciprocals and matrix instructions have to share a basic block for the group barriers below; behind the
        //  loop range's last chunk the product of the block's next chunk -- staged rows, zeros beyond the factor -- is computed and dropped)
        if (PIPE && ch + 1 < CPB) {
          first_product(cur, ch + 1, pnext);
          // (0x8: MFMA, 0x400: transcendental) 4 NT reciprocals over the 3 NB NT matrix instructions
#define HYB_MFMAS_THEN_RCP(m) \
  __builtin_amdgcn_sched_group_barrier(0x8, m, 0); \
  __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
          if constexpr (NB == 4) {
            HYB_MFMAS_THEN_RCP(3) HYB_MFMAS_THEN_RCP(3) HYB_MFMAS_THEN_RCP(3) HYB_MFMAS_THEN_RCP(3)
            HYB_MFMAS_THEN_RCP(3) HYB_MFMAS_THEN_RCP(3) HYB_MFMAS_THEN_RCP(3) HYB_MFMAS_THEN_RCP(3)
          } else if constexpr (NB == 3) {
            HYB_MFMAS_THEN_RCP(2) HYB_MFMAS_THEN_RCP(2) HYB_MFMAS_THEN_RCP(2) HYB_MFMAS_THEN_RCP(2)
            HYB_MFMAS_THEN_RCP(2) HYB_MFMAS_THEN_RCP(2) HYB_MFMAS_THEN_RCP(2) HYB_MFMAS_THEN_RCP(2)
          }
#undef HYB_MFMAS_THEN_RCP
        }
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2_t q2 = (f32x2_t){xr[ch & 3][t][r], xr[ch & 3][t][r + 1]} * (f32x2_t){q[t][r], q[t][r + 1]};
            q[t][r] = (!edge || dch + 4 * g + r < d1) ? q2.x : 0.0f;
            q[t][r + 1] = (!edge || dch + 4 * g + r + 1 < d1) ? q2.y : 0.0f;
          }
        }
        // second product: numerators of the signals 16 nb + 4g + r'
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const f32x4_t bn = *(const f32x4_t *)(cur + BFB + ch * CHT + (nb * 4 + g) * 256 + c16 * 16);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[r], q[t][r], acc[t][nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more && !(BN && !OBJ)) stage_write(nxt, d0 + 64 * (blk + 1));
    }
  }
  // acc[t][nb][r] = numerator of signal c = 16 nb + 4g + r at lane element l0 + 16t + c16

  if (OBJ || SSE) {  // workgroup sum in wave order (fixed order => reproducible)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_down(ssum, o, 64);
    __syncthreads();
    if (lane == 0) lds[wave] = ssum;
    __syncthreads();
    if (tid == 0) {
      double t = 0;
      for (int w = 0; w < 8; ++w) t += lds[w];
      ((double *)(arena + rdp->ossepart))[OBJ ? tile : bx] = t * weight * weight;
    }
    if (OBJ) return;
  }
  if (!gp->fused) {
    float *__restrict__ part = (float *)(arena + rdp->opart);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (lv[t]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * nb + 4 * g + r;
            if (c < kp) part[((int64_t)sp * L + lt[t]) * kp + c] = acc[t][nb][r];
          }
      }
    return;
  }
  double *den = lds;
  // this lane's old values of its four signals per block of sixteen (kp is a multiple of 4: whole quads), requested in front of
  // the denominators' barriers
  f32x4_t aold[NT][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      aold[t][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (lv[t] && 16 * nb + 4 * g < kp) aold[t][nb] = *(const f32x4_u *)(A + (int64_t)lt[t] * kp + 16 * nb + 4 * g);
    }
  __syncthreads();  // (every wave has left the staging buffers: the table goes there)
  if (tabpre) {
    double *tab = (double *)sb;
#pragma unroll
    for (int i = 0; i < NTAB; ++i)
      if (tid + 512 * i < ntab) tab[tid + 512 * i] = tabv[i];
    __syncthreads();
    if (tid < kp) {
      double sd = 0;
      for (int pp = 0; pp < PB; ++pp) sd += tab[pp * kp + tid];
      den[tid] = sd;
    }
  } else if (tid < kp) {
    den[tid] = nmfk_slot_sum<16>(sumB, kp, PB, tid);
  }
  __syncthreads();
  float *__restrict__ Anew = which == 0 ? (float *)(arena + NMFK_HOFF(*rdp, it + 1)) : (float *)(arena + rdp->oWt);
  double *sumA = (double *)(arena + (which == 0 ? rdp->osumH : rdp->osumW)) + (int64_t)tile * kp;
  double *red = den + NMFK_MAX_K;  // [8][kp]
  const float floorv = (gp->clampw > it + 1 && which == 1 && (it + 1) % 10 == 0) ? HYB_EPS : -__builtin_inff();
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c0 = 16 * nb + 4 * g;
    float vs[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < kp) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (lv[t]) {
          f32x4_t v4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (c0 + r < k) {
              v = aold[t][nb][r] * acc[t][nb][r] / (float)den[c0 + r];  // Mult:67 / Mult:70 order
              v = v < floorv ? floorv : v;                             // (NmfkStepArgs::clampw; a NaN stays)
            }
            v4[r] = v;
            vs[r] += v;
          }
          *(f32x4_u *)(Anew + (int64_t)lt[t] * kp + c0) = v4;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = (double)vs[r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (c16 == 0 && c0 + r < kp) red[wave * kp + c0 + r] = v;
    }
  }
  __syncthreads();
  if (tid < kp) {
    double t = red[tid];
    for (int w = 1; w < 8; ++w) t += red[w * kp + tid];
    sumA[tid] = tid < k ? t : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------
// The kernels.  ONE launch serves units of different kernel variants (NmfkRun::hyb = 4 / 8 / 16, see nmfk_hyb_variant):
// a workgroup belongs to one unit, so the switch is workgroup-uniform; registers and LDS are those of the widest variant
// (the same 4 waves per SIMD for all).  With a launch per variant the 96 + 128 + 256 factorizations of the bench sweep
// were three launches that each left CUs idle (H half-step: two workgroups per unit) and ended in their own tail;
// together they fill the chip (profiles/r03/variants_one_launch.txt).
// ------------------------------------------------------------------------------------------------------
template <int NT, int NW, int MODE>  // MODE 0: half-step, 1: objective, 2: half-step that leaves the objective of its inputs, 3: half-step of short loop ranges (no LAG)
__global__ __launch_bounds__(64 * (NW > 8 ? NW : 8), MODE == 1 ? 2 : 4) void hyb_step_kernel(char *arena, const float *__restrict__ Xa,
                                                       const float *__restrict__ Xt,
                                                       const NmfkRun *__restrict__ runs,
                                                       const NmfkState *__restrict__ state,
                                                       const NmfkStepArgs *__restrict__ gp, int it, int u0, double weight) {
  extern __shared__ double lds[];
  switch (runs[u0 + blockIdx.y].hyb) {
    case 4: hyb_step_body<4, MODE == 1 ? 0 : 1, NT, NW, MODE == 1, MODE == 2, MODE == 0 || MODE == 2>(arena, Xa, Xt, runs, state, gp, it, u0, weight, lds); break;
    case 8: hyb_step_body<8, MODE == 1 ? 0 : 2, NT, NW, MODE == 1, MODE == 2, MODE == 0 || MODE == 2>(arena, Xa, Xt, runs, state, gp, it, u0, weight, lds); break;
    default: hyb_step_body<16, 0, NT, NW, MODE == 1, MODE == 2, MODE == 0 || MODE == 2>(arena, Xa, Xt, runs, state, gp, it, u0, weight, lds); break;
  }
}

template <int NT, bool OBJ, bool RAG = false, bool SSE = false>
__global__ __launch_bounds__(64 * NMFK_HYB_RW) void hyb_res_kernel(char *arena, const float *__restrict__ Xt, const NmfkRun *__restrict__ runs,
                                                      const NmfkState *__restrict__ state, const NmfkStepArgs *__restrict__ gp,
                                                      int it, int u0, double weight, int ntile) {
  extern __shared__ double lds[];
  switch (runs[u0 + blockIdx.y].hyb) {
    case 4: hyb_res_body<4, OBJ ? 0 : 1, NT, OBJ, RAG, SSE>(arena, Xt, runs, state, gp, it, u0, weight, ntile, lds); break;
    case 8: hyb_res_body<8, OBJ ? 0 : 2, NT, OBJ, RAG, SSE>(arena, Xt, runs, state, gp, it, u0, weight, ntile, lds); break;
    default: hyb_res_body<16, 0, NT, OBJ, RAG, SSE>(arena, Xt, runs, state, gp, it, u0, weight, ntile, lds); break;
  }
}

}  // namespace

#ifndef NMFK_HYB_NT
#define NMFK_HYB_NT 2  // 16-wide lane tiles per wave
#endif
#ifndef NMFK_HYB_NW
#define NMFK_HYB_NW 8  // waves per workgroup (wsplit = 1): they share the staged chunks of the loop factor
#endif

int nmfk_hyb_lane_tile(int wsplit) { return 16 * NMFK_HYB_NT * (wsplit > 1 ? 1 : NMFK_HYB_NW); }

// kernel variant of a rank (NmfkRun::hyb): 4 = (KS 4, NS 1) for k <= 4, 8 = (KS 8, NS 2) for k <= 8, 16 = (KS 16, 16-signal
// numerators) for k <= 16.  (Three sets of 4x4x1 numerators for k = 9..12 were built and measured: at 128 registers per
// wave they spill, and in the resident form, where they fit, a launch of 256 factorizations of k = 12 took 0.473 ms
// against 0.470 ms with the 16-signal form -- that kernel is bound by instruction issue, not by the matrix pipe.)
int nmfk_hyb_variant(int k) { return k <= 4 ? 4 : k <= 8 ? 8 : 16; }

namespace {
// bytes of one staged chunk (split planes + transposed block) of the widest variant <= vmax
size_t hyb_chunk_bytes(int vmax) {
  size_t b = 0;
  for (int v : {4, 8, 16}) {
    if (v > vmax) continue;
    const size_t planes = 3 * (size_t)(v == 4 ? 1 : v / 8) * 256;
    b = std::max(b, planes + (v == 16 ? 4 * 256 : 4 * 320));
  }
  return b;
}
}  // namespace

int nmfk_hyb_resident_waves() { return NMFK_HYB_RW; }
// Resident form: bytes of LDS a workgroup needs for a loop dimension of D when the widest variant of the launch is vmax
// (0: not applicable -- shorter than one trip, or too long); any D: the kernel walks roundup64(D) steps
size_t nmfk_hyb_resident_lds(int vmax, int D) {
  if (D < 64) return 0;
  const size_t need = sizeof(double) * 18 * 16 + (size_t)(((D + 63) & ~63) >> 4) * hyb_chunk_bytes(vmax);
  return need <= 160 * 1024 ? need : 0;
}

// half-step of the `cnt` units [u0, u0 + cnt) (any mix of variants; vmax = the widest among them)
// objw > 0: the launch also leaves the objective of the factors it reads, scaled
// by objw^2, as nmfk_hyb_step_parts(a) partials per unit in NmfkRun::ossepart
void nmfk_launch_step_hyb_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int vmax, int u0, int cnt, hipStream_t s, double objw) {
  constexpr int NT = NMFK_HYB_NT, NW = NMFK_HYB_NW;
  if (a.res_wgs > 0) {  // resident form (the host has checked nmfk_hyb_resident_lds)
    static std::atomic<uint64_t> lds_ok[4] = {{0}, {0}, {0}, {0}};  // (more than 64 KB of dynamic LDS: per kernel and device)
    const dim3 grid(a.res_wgs, cnt), blk(64 * NMFK_HYB_RW);
    const size_t ldsb = nmfk_hyb_resident_lds(vmax, a.D);
#define NMFK_RES_LAUNCH(RAGV, SSEV, W)                                                                          \
  do {                                                                                                          \
    nmfk_allow_dynamic_lds((const void *)hyb_res_kernel<NT, false, RAGV, SSEV>, lds_ok[2 * RAGV + SSEV], 160 * 1024); \
    hipLaunchKernelGGL((hyb_res_kernel<NT, false, RAGV, SSEV>), grid, blk, ldsb, s, a.arena, a.Xtile, a.runs, a.state, dargs, a.it, u0, W, 0); \
  } while (0)
    const bool rag = (a.D & 63) != 0;
    if (objw > 0) {  // the launch also leaves the objective of the factors it reads: res_wgs partials per unit
      if (rag)
        NMFK_RES_LAUNCH(true, true, objw);
      else
        NMFK_RES_LAUNCH(false, true, objw);
    } else {
      if (rag)
        NMFK_RES_LAUNCH(true, false, 1.0);
      else
        NMFK_RES_LAUNCH(false, false, 1.0);
    }
#undef NMFK_RES_LAUNCH
    return;
  }
  const int ws = a.wsplit, nwaves = ws > 1 ? ws : NW;
  const int lpw = 16 * NT * (ws > 1 ? 1 : nwaves);
  const int ntile = (a.L + lpw - 1) / lpw;
  const dim3 grid(ntile * a.S, cnt), blk(64 * nwaves);
  const size_t cross = ws > 1 ? (size_t)(ws - 1) * NT * 4 * 64 * sizeof(float) : 0;
  // two staged blocks of NMFK_HYB_CPB chunks per workgroup (wsplit = 1) / of one chunk per wave (wsplit > 1): HybStage::STB
  const size_t chunkb = hyb_chunk_bytes(vmax);
  const size_t stage = ws > 1 ? (size_t)ws * 2 * chunkb : 2 * NMFK_HYB_CPB * chunkb;
  const size_t ldsb = sizeof(double) * 18 * 16 + std::max(cross, stage);
  if (objw > 0)
    hipLaunchKernelGGL((hyb_step_kernel<NT, NW, 2>), grid, blk, ldsb, s, a.arena, a.Xalt, a.Xtile, a.runs, a.state, dargs, a.it, u0, objw);
  else if (a.lag < 0 ? (a.D + a.S - 1) / a.S / (ws > 1 ? ws : 1) >= 512 : a.lag > 0)  // loop rows a wave walks: 32 chunks and more run the lagged form (hyb_step_body, LAG)
    hipLaunchKernelGGL((hyb_step_kernel<NT, NW, 0>), grid, blk, ldsb, s, a.arena, a.Xalt, a.Xtile, a.runs, a.state, dargs, a.it, u0, 1.0);
  else
    hipLaunchKernelGGL((hyb_step_kernel<NT, NW, 3>), grid, blk, ldsb, s, a.arena, a.Xalt, a.Xtile, a.runs, a.state, dargs, a.it, u0, 1.0);
}

// objective partials per unit a launch with these arguments leaves in its objective mode
int nmfk_hyb_step_parts(const NmfkStepArgs &a) {
  if (a.res_wgs > 0) return a.res_wgs;  // (the resident form: a partial per workgroup)
  const int lpw = nmfk_hyb_lane_tile(a.wsplit);
  return (a.L + lpw - 1) / lpw * a.S;
}

// tiled copy of X (element (l, d) at src[d + l*D]) for nmfk_launch_step_hyb_f32; out: roundup16(L) * roundup16(D) floats
void nmfk_launch_hyb_tile(const float *src, int L, int D, float *out, hipStream_t s) {
  const int64_t total = (int64_t)((L + 15) / 16) * ((D + 15) / 16) * 256;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (total + NMFK_TILE - 1) / NMFK_TILE));
  hipLaunchKernelGGL(hyb_tile_kernel, dim3(nb), dim3(NMFK_TILE), 0, s, src, L, D, out);
}

// monitored objective of the units [u0, u0 + cnt) of a group on the split-operand MFMA kernel (active units only):
// the half-step kernel in its objective mode.  w: the W half-step's arguments (dw: their device copy); hsel: parity
// of the H buffer that holds the current H
void nmfk_launch_hyb_sse(const NmfkStepArgs &w, const NmfkStepArgs *dw, double weight, int hsel, int vmax, int u0, int cnt,
                         hipStream_t s) {
  constexpr int NT = NMFK_HYB_NT, NW = 8;
  if (w.res_wgs > 0) {  // resident form (W orientation: the loop factor H sits in LDS), res_wgs partials per unit
    static std::atomic<uint64_t> lds_ok{0}, lds_okr{0};  // (more than 64 KB of dynamic LDS: per kernel and device)
    const int ntile = (w.L + NMFK_TILE - 1) / NMFK_TILE;  // partials check_a_kernel adds (sse_kernel's count)
    if ((w.D & 63) != 0) {
      nmfk_allow_dynamic_lds((const void *)hyb_res_kernel<NT, true, true>, lds_okr, 160 * 1024);
      hipLaunchKernelGGL((hyb_res_kernel<NT, true, true>), dim3(std::min(w.res_wgs, ntile), cnt), dim3(64 * NMFK_HYB_RW),
                         nmfk_hyb_resident_lds(vmax, w.D), s, w.arena, w.Xtile, w.runs, w.state, dw, hsel, u0, weight, ntile);
    } else {
      nmfk_allow_dynamic_lds((const void *)hyb_res_kernel<NT, true>, lds_ok, 160 * 1024);
      hipLaunchKernelGGL((hyb_res_kernel<NT, true>), dim3(std::min(w.res_wgs, ntile), cnt), dim3(64 * NMFK_HYB_RW),
                         nmfk_hyb_resident_lds(vmax, w.D), s, w.arena, w.Xtile, w.runs, w.state, dw, hsel, u0, weight, ntile);
    }
    return;
  }
  const int lpw = 16 * NT * NW;  // = NMFK_TILE: the partials line up with sse_kernel's
  const dim3 grid((w.L + lpw - 1) / lpw, cnt), blk(64 * NW);
  const size_t ldsb = sizeof(double) * 18 * 16 + 2 * NMFK_HYB_CPB * 3 * (size_t)(vmax <= 8 ? 1 : 2) * 256;  // two blocks of split planes
  hipLaunchKernelGGL((hyb_step_kernel<NT, NW, 1>), grid, blk, ldsb, s, w.arena, w.Xalt, w.Xtile, w.runs, w.state, dw, hsel, u0, weight);
}

// Wide ranks on the split-operand first product (wide2_step_kernel): kp <= 32 (KS = 32), kp = 40, 48 (KS = 48, round 4: padded to
// 64 signals those widths lost against the all-fp32 kernel -- k = 48: 7.39 vs 5.68 ms per iteration of 8 restarts at
// 65536 x 2048) and kp >= 56 (KS = 64).
int nmfk_wide2_ok(int kp) { return kp > 16 && kp <= 64; }
static int wide2_nb(int kp) { return kp <= 32 ? 2 : kp <= 48 ? 3 : 4; }
int nmfk_wide2_lane_tile() { return 16 * NMFK_HYB_NT * 8; }

void nmfk_launch_step_wide2_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt, hipStream_t s, double objw) {
  constexpr int NT = NMFK_HYB_NT;
  const int lpw = 16 * NT * 8, ntile = (a.L + lpw - 1) / lpw;
  const dim3 grid(ntile * a.S, cnt), blk(512);
  const int nb = wide2_nb(kp);
  // numerators on the bf16 pipe (NmfkStepArgs::bnum): by default at 48 and 64 signals -- measured at 65536 x 2048, 8 restarts, ms per iteration fp32 -> bf16
  // numerators: k = 64 3.58 -> 3.10, k = 48 2.83 -> 2.62, k = 40 2.81 -> 2.63; at 32 signals the form needs 164 registers instead of 128, i.e. one
  // workgroup per CU instead of two, and loses: k = 32 2.02 -> 2.31, k = 24 1.96 -> 2.21 (profiles/r06/wide_bn_ab.txt).  bnum = 2 forces it everywhere.
  const bool bn = a.bnum >= 2 || (a.bnum == 1 && nb >= 3);
  const size_t chunkb = 3 * (size_t)(2 * nb) * 256 + (size_t)nb * (bn ? HYB_BN_BLK : 1024);
  const size_t ldsb = sizeof(double) * 9 * NMFK_MAX_K + 2 * 4 * chunkb;
  // objw > 0: the launch also leaves the objective of the factors it reads (scaled by objw^2; grid.x partials per unit)
  // (every instantiation may need more than 64 KB of dynamic LDS: allowed per kernel and device)
#define NMFK_WIDE2_LAUNCH(NBV, MODE, BNV, W)                                                                  \
  do {                                                                                                        \
    static std::atomic<uint64_t> lds_ok{0};                                                                   \
    nmfk_allow_dynamic_lds((const void *)wide2_step_kernel<NBV, NT, MODE, BNV>, lds_ok, 160 * 1024);          \
    hipLaunchKernelGGL((wide2_step_kernel<NBV, NT, MODE, BNV>), grid, blk, ldsb, s, a.arena, a.Xtile, a.runs, a.state, dargs, a.it, u0, W); \
  } while (0)
#define NMFK_WIDE2_PICK(NBV)                    \
  do {                                          \
    if (objw > 0) {                             \
      if (bn)                                   \
        NMFK_WIDE2_LAUNCH(NBV, 2, true, objw);  \
      else                                      \
        NMFK_WIDE2_LAUNCH(NBV, 2, false, objw); \
    } else {                                    \
      if (bn)                                   \
        NMFK_WIDE2_LAUNCH(NBV, 0, true, 1.0);   \
      else                                      \
        NMFK_WIDE2_LAUNCH(NBV, 0, false, 1.0);  \
    }                                           \
  } while (0)
  if (nb == 2)
    NMFK_WIDE2_PICK(2);
  else if (nb == 3)
    NMFK_WIDE2_PICK(3);
  else
    NMFK_WIDE2_PICK(4);
#undef NMFK_WIDE2_PICK
#undef NMFK_WIDE2_LAUNCH
}

// monitored objective of wide-rank units (scalar weight, no missing data): the kernel above in its objective mode.
// w: the W half-step's arguments (dw: device copy); hsel: parity of the H buffer that holds the current H
void nmfk_launch_wide2_sse(const NmfkStepArgs &w, const NmfkStepArgs *dw, double weight, int hsel, int kp, int u0, int cnt, hipStream_t s) {
  constexpr int NT = NMFK_HYB_NT;
  const int lpw = 16 * NT * 8;  // = NMFK_TILE: the partials line up with sse_kernel's
  const dim3 grid((w.L + lpw - 1) / lpw, cnt), blk(512);
  const int nb = wide2_nb(kp);
  const size_t chunkb = 3 * (size_t)(2 * nb) * 256 + (size_t)nb * 1024;
  const size_t ldsb = sizeof(double) * 9 * NMFK_MAX_K + 2 * 4 * chunkb;
  if (nb == 2) {
    hipLaunchKernelGGL((wide2_step_kernel<2, NT, 1>), grid, blk, ldsb, s, w.arena, w.Xtile, w.runs, w.state, dw, hsel, u0, weight);
  } else if (nb == 3) {
    static std::atomic<uint64_t> lds_ok3{0};  // (more than 64 KB of dynamic LDS: per kernel and device)
    nmfk_allow_dynamic_lds((const void *)wide2_step_kernel<3, NT, 1>, lds_ok3, 160 * 1024);
    hipLaunchKernelGGL((wide2_step_kernel<3, NT, 1>), grid, blk, ldsb, s, w.arena, w.Xtile, w.runs, w.state, dw, hsel, u0, weight);
  } else {
    static std::atomic<uint64_t> lds_ok{0};  // (more than 64 KB of dynamic LDS: per kernel and device)
    nmfk_allow_dynamic_lds((const void *)wide2_step_kernel<4, NT, 1>, lds_ok, 160 * 1024);
    hipLaunchKernelGGL((wide2_step_kernel<4, NT, 1>), grid, blk, ldsb, s, w.arena, w.Xtile, w.runs, w.state, dw, hsel, u0, weight);
  }
}
