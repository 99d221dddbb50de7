// Issue-rate probe, part 3 (round 6): the chunk of the k = 9..16 half-step as an instruction stream in registers, today's form against
// the form with the NUMERATORS ON THE BF16 PIPE (the ratios split into three bf16 terms with plain vector instructions).
//   per pair of lane tiles and chunk of 16 loop steps
//   OLD   6 v_mfma_f32_16x16x32_bf16 (W*H of the next chunk) + 8 v_rcp_f32 | 4 v_pk_mul_f32 | 8 v_mfma_f32_16x16x4_f32
//   NEWA  6 bf16 MFMA + 8 rcp | 8 v_mul + split by v_and / v_sub (4 per value) + 12 v_perm_b32 | 6 bf16 MFMA
//   NEWD  6 bf16 MFMA + 8 rcp | 8 v_mul + 12 v_perm_b32 + split by v_dot2c_f32_bf16 (2 per value)  | 6 bf16 MFMA
// Also: v_dot2c_f32_bf16 / v_perm_b32 alone and n of them behind every bf16 MFMA (coissue2's table, two more columns), and whether the
// dot2c split is EXACT (q = h + m + l bit for bit, the same terms as the and/sub split) over random values.
// 16 waves per workgroup = 4 per SIMD, 256 workgroups, wall clock by HIP events.
// Build: hipcc -O3 --offload-arch=gfx950 -w coissue3.hip -o coissue3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int TRIPS = 4096;

__device__ __forceinline__ uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bfloat(uint32_t v) { return __builtin_bit_cast(float, v); }
// packed upper halves: low = a's, high = b's
__device__ __forceinline__ uint32_t pack_hi(float a, float b) { return __builtin_amdgcn_perm(fbits(b), fbits(a), 0x07060302u); }
__device__ __forceinline__ float dot2c(float acc, uint32_t pk, uint32_t c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, c), acc, false);
}

// three-term split of four ratios of a lane tile into the operand words [m01 m23 h01 h23 l01 l23] (truncation split, plain instructions)
template <bool DOT>
__device__ __forceinline__ void split4(const f32x4 q, uint32_t (&w)[6]) {
  float r1[4], r2[4];
  w[2] = pack_hi(q[0], q[1]);
  w[3] = pack_hi(q[2], q[3]);
  if (DOT) {
    r1[0] = dot2c(q[0], w[2], 0x0000bf80u);
    r1[1] = dot2c(q[1], w[2], 0xbf800000u);
    r1[2] = dot2c(q[2], w[3], 0x0000bf80u);
    r1[3] = dot2c(q[3], w[3], 0xbf800000u);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) r1[i] = q[i] - bfloat(fbits(q[i]) & 0xffff0000u);
  }
  w[0] = pack_hi(r1[0], r1[1]);
  w[1] = pack_hi(r1[2], r1[3]);
  if (DOT) {
    r2[0] = dot2c(r1[0], w[0], 0x0000bf80u);
    r2[1] = dot2c(r1[1], w[0], 0xbf800000u);
    r2[2] = dot2c(r1[2], w[1], 0x0000bf80u);
    r2[3] = dot2c(r1[3], w[1], 0xbf800000u);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) r2[i] = r1[i] - bfloat(fbits(r1[i]) & 0xffff0000u);
  }
  w[4] = pack_hi(r2[0], r2[1]);
  w[5] = pack_hi(r2[2], r2[3]);
}

// exactness of the split: out[i] = {h, m, l packed as (a:b) words are not needed -- the three terms of value i as floats}
template <bool DOT>
__global__ void split_check(const float *in, float *out, int n) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 q = {in[i], in[i + 1], in[i + 2], in[i + 3]};
  uint32_t w[6];
  split4<DOT>(q, w);
  for (int e = 0; e < 4; ++e) {
    const int word = e >> 1, hi = e & 1;
    auto term = [&](uint32_t ww) { return bfloat(hi ? (ww & 0xffff0000u) : (ww << 16)); };
    out[3 * (i + e) + 0] = term(w[2 + word]);
    out[3 * (i + e) + 1] = term(w[0 + word]);
    out[3 * (i + e) + 2] = term(w[4 + word]);
  }
}

// V: 0 = OLD, 1 = NEWA, 2 = NEWD; LDS: operands of both products come from LDS (as in the resident form) instead of sitting in registers
template <int V, bool LDS>
__global__ __launch_bounds__(1024) void stream(const float *src, float seed, float *sink) {
  extern __shared__ char dynlds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 1024) ((float *)dynlds)[i] = seed + (i & 255) * 1e-3f;
  __syncthreads();
  bf16x8 bop[2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 3; ++j) bop[t][j] = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u + t, 0x3f803f80u + j, 0x3f003f00u, 0x3f003f00u});
  u32x4 avn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) avn[j] = (u32x4){0x3f803f80u, 0x3e803e80u + j, 0x3f003f00u, 0x3f803f80u};
  f32x4 x[2] = {{seed, seed + 1, seed + 2, seed + 3}, {seed + 4, seed + 5, seed + 6, seed + 7}};
  f32x4 accs[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, pc[2] = {{seed, seed, seed, seed}, {seed, seed, seed, seed}};
  f32x4 bn = {seed, seed * 0.5f, seed * 0.25f, seed * 0.125f};
  u32x4 a2[3] = {{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, {0x38003800u, 0x38003800u, 0x3f803f80u, 0x3f803f80u}};
  const char *lbase = dynlds + ((wave & 3) * 64 + lane) * 16;
#pragma unroll 1
  for (int tr = 0; tr < TRIPS; ++tr) {
    if (LDS) {
      if (V == 0) bn = *(const f32x4 *)(lbase + 8192 + (tr & 7) * 4096);
      else {
#pragma unroll
        for (int j = 0; j < 3; ++j) a2[j] = *(const u32x4 *)(lbase + 8192 + j * 4096 + (tr & 3) * 12288);
      }
    }
    // first product of the next chunk, the reciprocals of this chunk's beside it
    f32x4 pn[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) pn[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, avn[j]), bop[t][j], pn[t], 0, 0, 0);
    f32x4 q[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) q[t][r] = __builtin_amdgcn_rcpf(pc[t][r]);
#define MFMA_THEN_RCP(n) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x400, n, 0);
    MFMA_THEN_RCP(1) MFMA_THEN_RCP(2) MFMA_THEN_RCP(1) MFMA_THEN_RCP(1) MFMA_THEN_RCP(2) MFMA_THEN_RCP(1)
    __builtin_amdgcn_sched_barrier(0);
    if (LDS) {
#pragma unroll
      for (int j = 0; j < 3; ++j) avn[j] = *(const u32x4 *)(lbase + j * 4096 + (tr & 1) * 256);
    }
    if (V == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 q2 = (f32x2){x[t][r], x[t][r + 1]} * (f32x2){q[t][r], q[t][r + 1]};
          q[t][r] = q2.x;
          q[t][r + 1] = q2.y;
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[r], q[t][r], accs[t], 0, 0, 0);
    } else {
      uint32_t w[2][6];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) q[t][r] = x[t][r] * q[t][r];
        split4<V == 2>(q[t], w[t]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        typedef uint32_t u32x6 __attribute__((ext_vector_type(6)));
        const u32x6 w6 = {w[t][0], w[t][1], w[t][2], w[t][3], w[t][4], w[t][5]};
        const bf16x8 qmh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(w6, w6, 0, 1, 2, 3));
        const bf16x8 qhl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(w6, w6, 2, 3, 4, 5));
        accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2[0]), qmh, accs[t], 0, 0, 0);
        accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2[1]), qmh, accs[t], 0, 0, 0);
        accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2[2]), qhl, accs[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) pc[t] = pn[t] + x[t];  // (keeps W*H of the next chunk away from 0 and the chain alive)
    __builtin_amdgcn_sched_barrier(0);
  }
  sink[(size_t)blockIdx.x * 1024 + threadIdx.x] = accs[0][0] + accs[0][1] + accs[1][2] + accs[1][3] + pc[0][0];
}

// n vector instructions of kind B behind every bf16 MFMA (coissue2's table): 0 = v_dot2c_f32_bf16 (literal), 1 = v_perm_b32, 2 = v_dot2c (VGPR constant)
template <int B, int N>
__global__ __launch_bounds__(1024) void beside(float seed, float *sink) {
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  bf16x8 ab = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), bb = ab;
  float v[8];
  uint32_t u[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = seed + j; u[j] = fbits(seed) + j; }
  uint32_t cst = 0x0000bf80u;
  asm volatile("" : "+v"(cst));
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j & 3], 0, 0, 0);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const int s = (j * N + q) & 7;
        if (B == 0) asm volatile("v_dot2c_f32_bf16 %0, 0xbf80, %1" : "+v"(v[s]) : "v"(u[s]));
        if (B == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[s]) : "v"(u[(s + 1) & 7]), "s"(0x07060302u));
        if (B == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[s]) : "v"(cst), "v"(u[s]));
      }
    }
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j] + bfloat(u[j]);
  sink[(size_t)blockIdx.x * 1024 + threadIdx.x] = s;
}

template <typename F>
float timed(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int V, bool LDS>
void run_stream(const char *name, const float *src, float *sink) {
  (void)hipFuncSetAttribute((const void *)stream<V, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const float ms = timed([&] { hipLaunchKernelGGL((stream<V, LDS>), dim3(256), dim3(1024), 65536, 0, src, 1.5f, sink); });
  // per SIMD: 4 waves x TRIPS chunk-pairs; cycles at 2.25 GHz (coissue2's calibration: 14.2 ns = 32 cycles)
  const double ns = ms * 1e6 / TRIPS;
  printf("%-6s operands %-9s | kernel %8.1f us | %7.1f ns per chunk of a wave = %6.1f cycles per SIMD and chunk (4 waves per SIMD)\n", name,
         LDS ? "from LDS" : "registers", ms * 1e3, ns, ns / 4 * 2.25);
}
template <int B, int N>
void run_beside(const char *name, float *sink) {
  const float ms = timed([&] { hipLaunchKernelGGL((beside<B, N>), dim3(256), dim3(1024), 0, 0, 1.5f, sink); });
  const double ns = ms * 1e6 / (TRIPS * 8);
  printf("v_mfma_f32_16x16x32_bf16 + %d x %-28s | %6.1f ns per matrix instruction of a wave = %5.1f cycles per SIMD\n", N, name, ns, ns / 4 * 2.25);
}

int main() {
  float *sink, *src;
  (void)hipMalloc(&sink, sizeof(float) * 1024 * 256);
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  // ---- exactness of the two splits
  const int n = 1 << 20;
  std::vector<float> h(n), oa(3 * n), od(3 * n);
  uint64_t st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    uint32_t b = (uint32_t)(st >> 32);
    if (i & 1) b = (b & 0x007fffffu) | ((100u + (b >> 24) % 60u) << 23);  // ratios of ordinary size: 2^-27 .. 2^32
    else b = (b & 0x7fffffffu) % 0x7f000000u;                              // any positive finite value, tiny ones included
    memcpy(&h[i], &b, 4);
  }
  float *din, *dout;
  (void)hipMalloc(&din, 4 * n);
  (void)hipMalloc(&dout, 12 * n);
  (void)hipMemcpy(din, h.data(), 4 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((split_check<false>), dim3(n / 4 / 256), dim3(256), 0, 0, din, dout, n);
  (void)hipMemcpy(oa.data(), dout, 12 * n, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL((split_check<true>), dim3(n / 4 / 256), dim3(256), 0, 0, din, dout, n);
  (void)hipMemcpy(od.data(), dout, 12 * n, hipMemcpyDeviceToHost);
  long bad_sum_a = 0, bad_sum_d = 0, differ = 0, bad_small = 0;
  for (int i = 0; i < n; ++i) {
    const double sa = (double)oa[3 * i] + oa[3 * i + 1] + oa[3 * i + 2], sd = (double)od[3 * i] + od[3 * i + 1] + od[3 * i + 2];
    const bool small = h[i] < 1e-30f;
    if (sa != (double)h[i]) { if (small) ++bad_small; else ++bad_sum_a; }
    if (sd != (double)h[i] && !small) ++bad_sum_d;
    if (memcmp(&oa[3 * i], &od[3 * i], 12) != 0 && !small) ++differ;
  }
  printf("split exactness over %d values: and/sub split h+m+l != q: %ld (values >= 1e-30; below: %ld)   dot2c split h+m+l != q: %ld   terms differ between the two: %ld\n", n,
         bad_sum_a, bad_small, bad_sum_d, differ);
  // ---- issue costs of the new instructions beside the bf16 matrix instruction
  run_beside<0, 0>("(nothing)", sink);
  run_beside<0, 1>("v_dot2c_f32_bf16 (literal)", sink);
  run_beside<0, 2>("v_dot2c_f32_bf16 (literal)", sink);
  run_beside<0, 4>("v_dot2c_f32_bf16 (literal)", sink);
  run_beside<2, 2>("v_dot2c_f32_bf16 (vgpr const)", sink);
  run_beside<2, 4>("v_dot2c_f32_bf16 (vgpr const)", sink);
  run_beside<1, 1>("v_perm_b32", sink);
  run_beside<1, 2>("v_perm_b32", sink);
  run_beside<1, 4>("v_perm_b32", sink);
  // ---- the chunk streams
  for (int rep = 0; rep < 2; ++rep) {
    run_stream<0, false>("OLD", src, sink);
    run_stream<1, false>("NEWA", src, sink);
    run_stream<2, false>("NEWD", src, sink);
    run_stream<0, true>("OLD", src, sink);
    run_stream<1, true>("NEWA", src, sink);
    run_stream<2, true>("NEWD", src, sink);
  }
  return 0;
}
