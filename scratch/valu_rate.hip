// Measures VALU issue rates on gfx950: plain v_fma_f32 vs v_pk_fma_f32, VGPR vs SGPR operands, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* sc, int iters) {
  float s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3];
  float a[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (v2f){a[i], a[i] + 1}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (MODE == 0) {  // plain fma, vgpr operands, 8 independent chains
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], a[(i + 1) & 7], 0.5f);
      } else if (MODE == 1) {  // plain fma with SGPR operand
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], (i & 1) ? s0 : s1, a[i]);
      } else if (MODE == 2) {  // pk fma vgpr operands
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], p[(i + 1) & 7], p[i]);
      } else if (MODE == 3) {  // pk fma with SGPR pair operand
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (i & 1) ? (v2f){s0, s1} : (v2f){s2, s3}, p[i]);
      } else if (MODE == 4) {  // pk fma: sgpr pair * broadcast vgpr (acc pattern)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma((i & 1) ? (v2f){s0, s1} : (v2f){s2, s3}, (v2f)(a[i & 3]), p[i]);
      } else if (MODE == 5) {  // rcp
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_rcpf(a[i]);
      }
    }
  }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> double run(int wgs_per_cu, int iters) {
  float *out, *sc; hipMalloc(&out, 4 << 20); hipMalloc(&sc, 64); hipMemset(sc, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, sc, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, sc, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: wgs_per_cu waves per SIMD (256 thr = 4 waves = 1 per SIMD) * iters * REP
  double instr = (double)wgs_per_cu * iters * REP;
  double cycles = ms * 1e-3 * 2.4e9;
  hipFree(out); hipFree(sc);
  return cycles / instr;  // cycles (at nominal 2.4 GHz) per wave-instruction per SIMD
}
int main() {
  const char* names[] = {"v_fma vgpr", "v_fma sgpr", "pk_fma vgpr", "pk_fma sgprpair", "pk_fma sgpr*bcast", "v_rcp"};
  for (int w : {1, 2, 4, 8}) {
    double r[6] = {run<0>(w, 4000), run<1>(w, 4000), run<2>(w, 4000), run<3>(w, 4000), run<4>(w, 4000), run<5>(w, 4000)};
    printf("waves/SIMD=%d:", w);
    for (int i = 0; i < 6; ++i) printf("  %s=%.2f", names[i], r[i]);
    printf("  (cycles@2.4GHz per wave-instr per SIMD)\n");
  }
}
