// Do fp32 MFMA (v_mfma_f32_16x16x4_f32) and fp32 packed VALU FMAs overlap on one SIMD of gfx950?
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 a v_pk_fma_f32 loop.
// mode 0: both, mode 1: MFMA waves only, mode 2: VALU waves only.  Prints cycles per instruction for each kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
__global__ __launch_bounds__(512) void k(int mode, float *out, long long *cyc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_mfma = wave < 4;
  if ((mode == 1 && !is_mfma) || (mode == 2 && is_mfma)) return;
  float a = lane * 0.001f, b = 1.0f + lane * 1e-6f;
  long long t0, t1;
  if (is_mfma) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_IT; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f32x2 x0 = {a, b}, x1 = {b, a}, x2 = {a, a}, x3 = {b, b}, x4 = x0, x5 = x1, x6 = x2, x7 = x3;
    const f32x2 m = {0.999f, 1.001f}, ad = {1e-3f, 2e-3f};
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_IT; ++i) {
      x0 = __builtin_elementwise_fma(x0, m, ad); x1 = __builtin_elementwise_fma(x1, m, ad);
      x2 = __builtin_elementwise_fma(x2, m, ad); x3 = __builtin_elementwise_fma(x3, m, ad);
      x4 = __builtin_elementwise_fma(x4, m, ad); x5 = __builtin_elementwise_fma(x5, m, ad);
      x6 = __builtin_elementwise_fma(x6, m, ad); x7 = __builtin_elementwise_fma(x7, m, ad);
    }
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = x0[0] + x1[1] + x2[0] + x3[1] + x4[0] + x5[1] + x6[0] + x7[1];
  }
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float *out; long long *cyc;
  const int nb = 256;
  hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8 * 8);
  long long h[nb * 8];
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(cyc, 0, nb * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, mode, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, v = 0; int nm = 0, nv = 0;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) { if (!h[b*8+w]) continue; if (w < 4) { m += h[b*8+w]; nm++; } else { v += h[b*8+w]; nv++; } }
    // s_memtime counts at 100 MHz-derived constant clock? report raw ticks per instruction
    printf("mode %d: mfma ticks/instr %.2f (n=%d)  pk_fma ticks/instr %.2f (n=%d)\n", mode, nm ? m / nm / (4.0 * N_IT) : 0, nm, nv ? v / nv / (8.0 * N_IT) : 0, nv);
  }
  return 0;
}
