// MFMA issue-rate check for the structure of mfma_step_kernel: per "chunk" 16 P-MFMAs (4 chains x 4 deep) and
// 16 accumulate-MFMAs whose B operand is the P result (optionally through v_rcp).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float afrag[4][4], bP[4], bN[4];
  for (int t = 0; t < 4; ++t) for (int s = 0; s < 4; ++s) afrag[t][s] = 0.001f * (lane + t + s);
  for (int s = 0; s < 4; ++s) { bP[s] = 0.5f + 0.01f * s; bN[s] = 0.25f + 0.01f * s; }
  f32x4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 p[4];
    for (int t = 0; t < 4; ++t) p[t] = (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) p[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[s], afrag[t][s], p[t], 0, 0, 0);
    f32x4 q[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) q[t][r] = MODE == 0 ? p[t][r] : __builtin_amdgcn_rcpf(p[t][r]) * bN[r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bN[r], q[t][r], acc[t], 0, 0, 0);
    bP[it & 3] += 1e-6f;
  }
  float r = 0;
  for (int t = 0; t < 4; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(int wgs_per_cu, int iters) {
  float* out; hipMalloc(&out, 4 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  double chunks_per_simd = (double)wgs_per_cu * iters;  // 1 wave per SIMD per WG
  printf("mode %d waves/SIMD %d: %.0f cycles@2.4GHz per chunk per SIMD (MFMA-bound: 1024), %.1f TF\n", MODE, wgs_per_cu,
         ms * 1e-3 * 2.4e9 / chunks_per_simd, 1024.0 * 256 * wgs_per_cu * iters * 32 * 2048 / (ms * 1e-3) / 1e12 / 256);
  hipFree(out);
}
int main() { for (int w : {1, 2, 3, 4}) { run<0>(w, 4000); run<1>(w, 4000); } }
