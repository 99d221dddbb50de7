// Round 3 microbenchmark: what does one chunk (2 tiles of 16 lane elements x 16 loop steps) of a half-step cost per SIMD
// when the numerators N += B'Q run as v_mfma_f32_4x4x1_16B_f32 blocks (4 signals x 4 lane elements x 1 loop step, 16
// blocks per instruction: no padding of the rank to 16) instead of v_mfma_f32_16x16x4_f32, and the first product needs
// only ceil(6k/32) bf16 MFMAs?  All operands in registers.  Also checks the 4x4x1 operand/result layout.
//   NBF  bf16 MFMAs per tile (first product: 1 for k <= 5, 2 for k <= 10, 3 for k <= 16)
//   N44  4x4x1 MFMAs per tile (4 * ceil(k/4)), or 0
//   N164 16x16x4 MFMAs per tile (4 = the shipped design), or 0
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define N_IT 2048

__global__ void layout_check(float *out) {
  const int lane = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 100.0f * (lane + 1), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

template <int NBF, int N44, int N164, bool VALU>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, const float *in) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NS = N44 / 4 > 0 ? N44 / 4 : 1;
  bf16x8 av[3], bop[2][3];
  for (int j = 0; j < 3; ++j) {
    u32x4 w = {0x3f803f80u + lane + j, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + j};
    av[j] = __builtin_bit_cast(bf16x8, w);
    for (int t = 0; t < 2; ++t) bop[t][j] = __builtin_bit_cast(bf16x8, w);
  }
  f32x4 acc[2][NS];
  for (int t = 0; t < 2; ++t)
    for (int s = 0; s < NS; ++s) acc[t][s] = (f32x4){0, 0, 0, 0};
  f32x4 x[2], bn[NS];
  for (int t = 0; t < 2; ++t) x[t] = (f32x4){in[lane], in[lane + 64], in[lane + 128], in[lane + 192 + t]};
  for (int s = 0; s < NS; ++s) bn[s] = x[0] + (float)s;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_IT; ++i) {
    f32x4 p[2] = {{1e-3f, 1e-3f, 1e-3f, 1e-3f}, {1e-3f, 1e-3f, 1e-3f, 1e-3f}};
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(av[j]));
    for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(x[t]));
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(bn[s]));
#pragma unroll
    for (int j = 0; j < NBF; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[j], bop[t][j], p[t], 0, 0, 0);
    f32x4 q[2];
    if (VALU) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 rc = {__builtin_amdgcn_rcpf(p[t][r]), __builtin_amdgcn_rcpf(p[t][r + 1])};
          const f32x2 q2 = (f32x2){x[t][r], x[t][r + 1]} * rc;
          q[t][r] = q2.x;
          q[t][r + 1] = q2.y;
        }
    } else {
      for (int t = 0; t < 2; ++t) q[t] = p[t];
    }
    if (N44 > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < N44 / 4; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(bn[s][r], q[t][r], acc[t][s], 0, 0, 0);
    }
    if (N164 > 0) {
#pragma unroll
      for (int r = 0; r < N164; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[0][r & 3], q[t][r & 3], acc[t][0], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int t = 0; t < 2; ++t)
    for (int s = 0; s < NS; ++s) r += acc[t][s][0] + acc[t][s][3];
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NBF, int N44, int N164, bool VALU>
void run(float *out, long long *cyc, const float *in, const char *name, int wgs_per_cu) {
  const int nb = 256 * wgs_per_cu;  // workgroups of 8 waves: 2 waves per SIMD each
  static long long h[1024 * 8];
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NBF, N44, N164, VALU>), dim3(nb), dim3(512), 0, 0, out, cyc, in);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NBF, N44, N164, VALU>), dim3(nb), dim3(512), 0, 0, out, cyc, in);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, sizeof(long long) * nb * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nb * 8; ++i) s += h[i];
  const int wps = 2 * wgs_per_cu;
  // matrix-pipe cycles per chunk: 16 per bf16 MFMA, 8 per 4x4x1, 32 per 16x16x4; two tiles
  const double ideal = 2.0 * (16.0 * NBF + 8.0 * N44 + 32.0 * N164);
  printf("%-34s %d waves/SIMD: memtime ticks per chunk per SIMD %8.1f   wall ns per chunk per SIMD %7.1f (= %6.1f cyc @2.4GHz)   matrix-pipe ideal %5.0f\n",
         name, wps, s / (nb * 8) / N_IT / wps, (double)ms * 1e6 / N_IT / wps,
         (double)ms * 1e6 / N_IT / wps * 2.4, ideal);
}

int main() {
  float *out, *in;
  long long *cyc;
  hipMalloc(&out, 1024 * 512 * 4);
  hipMalloc(&cyc, 1024 * 8 * 8);
  hipMalloc(&in, 4096);
  float hin[1024];
  for (int i = 0; i < 1024; ++i) hin[i] = 0.5f + 0.001f * i;
  hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
  {
    hipLaunchKernelGGL(layout_check, dim3(1), dim3(64), 0, 0, out);
    float h[256];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 4; ++r) {
        const int b = lane / 4;
        const float want = (float)(4 * b + r + 1) * 100.0f * (lane + 1);  // D_b[r][j] = A_b[r] * B_b[j], lane = 4b + j
        if (h[lane * 4 + r] != want) ++bad;
      }
    printf("4x4x1 layout check (D[vgpr r][lane 4b+j] = A[lane 4b+r] * B[lane 4b+j]): %s (%d mismatches); lane 5: %g %g %g %g\n",
           bad ? "DIFFERENT" : "as assumed", bad, h[20], h[21], h[22], h[23]);
  }
  for (int w : {2, 1}) {
    run<3, 0, 4, true>(out, cyc, in, "shipped KS=16: 3 bf16 + 4 f32 16x16x4", w);
    run<2, 0, 4, true>(out, cyc, in, "shipped KS=8:  2 bf16 + 4 f32 16x16x4", w);
    run<3, 16, 0, true>(out, cyc, in, "k 13-16: 3 bf16 + 16 4x4x1", w);
    run<3, 12, 0, true>(out, cyc, in, "k 11-12: 3 bf16 + 12 4x4x1", w);
    run<2, 12, 0, true>(out, cyc, in, "k 9-10:  2 bf16 + 12 4x4x1", w);
    run<2, 8, 0, true>(out, cyc, in, "k 6-8:   2 bf16 + 8 4x4x1", w);
    run<1, 8, 0, true>(out, cyc, in, "k 5:     1 bf16 + 8 4x4x1", w);
    run<1, 4, 0, true>(out, cyc, in, "k 2-4:   1 bf16 + 4 4x4x1", w);
    run<3, 16, 0, false>(out, cyc, in, "  (no VALU) 3 bf16 + 16 4x4x1", w);
    run<1, 4, 0, false>(out, cyc, in, "  (no VALU) 1 bf16 + 4 4x4x1", w);
    run<0, 16, 0, false>(out, cyc, in, "  (no VALU) 16 4x4x1 only", w);
    run<0, 0, 4, false>(out, cyc, in, "  (no VALU) 4 f32 16x16x4 only", w);
  }
  return 0;
}
