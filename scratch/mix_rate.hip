// What does one chunk of the split-operand MFMA half-step cost per SIMD on gfx950, with 4 waves per SIMD and all
// operands in registers?  Design A (round 1): 6 bf16 MFMAs (W*H) + 8 v_rcp + 8 v_mul + 8 fp32 MFMAs (numerators).
// Design B: 12 bf16 MFMAs + 8 v_rcp + 4 v_pk_mul + three-term bf16 split of the 8 ratios (numerators on the bf16 pipe).
// Prints SIMD cycles per chunk (2 tiles of 16 x 16) for A, B, and their MFMA-only / VALU-only parts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define N_IT 2048
__device__ __forceinline__ unsigned cvtpk(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
template <int MODE>  // 0: A full, 1: A mfma only, 2: A valu only, 3: B full, 4: B mfma only, 5: B valu only
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, const float *in) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 av[3], bop[2][3], nb[3];
  for (int j = 0; j < 3; ++j) {
    u32x4 w = {0x3f803f80u + lane + j, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + j};
    av[j] = __builtin_bit_cast(bf16x8, w);
    nb[j] = __builtin_bit_cast(bf16x8, w);
    for (int t = 0; t < 2; ++t) bop[t][j] = __builtin_bit_cast(bf16x8, w);
  }
  f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 x[2], bn;
  for (int t = 0; t < 2; ++t) x[t] = (f32x4){in[lane], in[lane + 64], in[lane + 128], in[lane + 192 + t]};
  bn = x[0];
  f32x4 vsum = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_IT; ++i) {
    f32x4 p[2] = {{1e-3f, 1e-3f, 1e-3f, 1e-3f}, {1e-3f, 1e-3f, 1e-3f, 1e-3f}};
    // opaque to the optimiser: the operands "change" every iteration (nothing is hoisted out of the loop)
    for (int j = 0; j < 3; ++j) { asm volatile("" : "+v"(av[j])); asm volatile("" : "+v"(nb[j])); }
    for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(x[t]));
    asm volatile("" : "+v"(bn));
    if (MODE != 2 && MODE != 5) {
      for (int j = 0; j < 3; ++j)
        for (int t = 0; t < 2; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[j], bop[t][j], p[t], 0, 0, 0);
    } else {
      for (int t = 0; t < 2; ++t) p[t] += x[t] + vsum;
    }
    f32x4 q[2];
    if (MODE != 1 && MODE != 4) {
      for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 4; ++r) q[t][r] = x[t][r] * __builtin_amdgcn_rcpf(p[t][r]);
    } else {
      for (int t = 0; t < 2; ++t) q[t] = p[t];
    }
    if (MODE < 3) {  // design A: numerators on the fp32 matrix pipe
      if (MODE != 2) {
        for (int r = 0; r < 4; ++r)
          for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[r], q[t][r], acc[t], 0, 0, 0);
      } else {
        vsum += q[0] + q[1];
      }
    } else {  // design B: three-term split of the ratios, numerators on the bf16 matrix pipe
      for (int t = 0; t < 2; ++t) {
        u32x4 b1, b3;
        if (MODE != 4) {
          unsigned h0 = cvtpk(q[t][0], q[t][1]), h1 = cvtpk(q[t][2], q[t][3]);
          f32x4 r1 = {q[t][0] - __builtin_bit_cast(float, h0 << 16), q[t][1] - __builtin_bit_cast(float, h0 & 0xffff0000u),
                      q[t][2] - __builtin_bit_cast(float, h1 << 16), q[t][3] - __builtin_bit_cast(float, h1 & 0xffff0000u)};
          unsigned m0 = cvtpk(r1[0], r1[1]), m1 = cvtpk(r1[2], r1[3]);
          f32x4 r2 = {r1[0] - __builtin_bit_cast(float, m0 << 16), r1[1] - __builtin_bit_cast(float, m0 & 0xffff0000u),
                      r1[2] - __builtin_bit_cast(float, m1 << 16), r1[3] - __builtin_bit_cast(float, m1 & 0xffff0000u)};
          unsigned l0 = cvtpk(r2[0], r2[1]), l1 = cvtpk(r2[2], r2[3]);
          b1 = (u32x4){h0, h1, m0, m1};
          b3 = (u32x4){h0, h1, l0, l1};
        } else {
          b1 = __builtin_bit_cast(u32x4, q[t]);
          b3 = b1;
        }
        if (MODE != 5) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nb[0], __builtin_bit_cast(bf16x8, b1), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nb[1], __builtin_bit_cast(bf16x8, b1), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nb[2], __builtin_bit_cast(bf16x8, b3), acc[t], 0, 0, 0);
        } else {
          vsum += __builtin_bit_cast(f32x4, b1) + __builtin_bit_cast(f32x4, b3);
        }
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + vsum[0] + vsum[3];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE>
void run(float *out, long long *cyc, const float *in, const char *name) {
  const int nb = 512;  // 2 workgroups of 8 waves per CU = 4 waves per SIMD
  static long long h[512 * 8];
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(512), 0, 0, out, cyc, in);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nb * 8; ++i) s += h[i];
  // s_memtime ticks at the constant 100 MHz reference; report ticks and let the caller scale -- all modes share it
  printf("%-28s wave ticks per chunk %.2f  => SIMD ticks per chunk (4 waves) %.2f\n", name, s / (nb * 8) / N_IT, s / (nb * 8) / N_IT / 4);
}
int main() {
  float *out, *in; long long *cyc;
  hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8 * 8); hipMalloc(&in, 4096);
  float hin[1024];
  for (int i = 0; i < 1024; ++i) hin[i] = 0.5f + 0.001f * i;
  hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
  run<0>(out, cyc, in, "A full");
  run<1>(out, cyc, in, "A mfma only (6 bf16+8 f32)");
  run<2>(out, cyc, in, "A valu only");
  run<3>(out, cyc, in, "B full");
  run<4>(out, cyc, in, "B mfma only (12 bf16)");
  run<5>(out, cyc, in, "B valu only");
  return 0;
}
