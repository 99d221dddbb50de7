// Issue-rate probe (round 5): what does a wave's vector work cost while ANOTHER wave of the same SIMD keeps the matrix pipe busy?
// A workgroup of 8 waves on every CU: the waves 0..3 (one per SIMD) run back-to-back matrix instructions of form A, the waves 4..7
// back-to-back independent vector instructions of form B; each role is timed alone (the other role leaves at once) and together.
// Output: cycles per instruction of both roles.  Build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TRIPS = 2048, UNROLL = 8;  // instructions per role = TRIPS * UNROLL

template <int A>
__device__ __forceinline__ void role_a(float seed, float *sink) {
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  float a = seed, b = seed * 0.5f;
  bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {8, 7, 6, 5, 4, 3, 2, 1};
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      if (A == 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
      if (A == 1) acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j & 3], 0, 0, 0);
      if (A == 2) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j & 3], 0, 0, 0);
    }
  }
  sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int B>
__device__ __forceinline__ void role_b(float seed, float *sink, float *lds) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = seed + j;
  f32x2 p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = f32x2{seed + j, seed - j};
  const f32x2 c = {1.0001f, 0.9999f};
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      if (B == 0) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j]));
      if (B == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(c));
      if (B == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c[0]));
      if (B == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[j]) : "v"(c[0]));
      if (B == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c[0]));
      if (B == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(c));
      if (B == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c[0]));
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j] + p[j][0] + p[j][1];
  sink[threadIdx.x] = s;
}

// mode: 1 = role A only, 2 = role B only, 3 = both
template <int A, int B>
__global__ __launch_bounds__(512) void probe(int mode, float seed, float *sink, long long *cyc) {
  __shared__ float lds[64];
  const int wave = threadIdx.x >> 6;
  const bool isA = wave < 4;
  if ((isA && !(mode & 1)) || (!isA && !(mode & 2))) return;
  float *mysink = sink + (size_t)blockIdx.x * 512;
  const long long t0 = clock64();
  if (isA) role_a<A>(seed, mysink); else role_b<B>(seed, mysink, lds);
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int A, int B>
void run(const char *na, const char *nb, float *sink, long long *cyc, int blocks) {
  double res[4] = {0, 0, 0, 0};  // A alone, B alone, A together, B together
  for (int mode = 1; mode <= 3; ++mode) {
    hipMemset(cyc, 0, sizeof(long long) * blocks * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<A, B>), dim3(blocks), dim3(512), 0, 0, mode, 1.5f, sink, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
    const double per = (double)TRIPS * UNROLL * blocks * 4;
    if (mode == 1) res[0] = sa / per;
    if (mode == 2) res[1] = sb / per;
    if (mode == 3) { res[2] = sa / per; res[3] = sb / per; }
  }
  printf("%-28s beside %-22s | alone: %6.2f / %6.2f   together: %6.2f / %6.2f cycles per instruction (matrix / vector)\n", na, nb, res[0], res[1], res[2], res[3]);
}

int main() {
  int blocks = 256;
  float *sink; long long *cyc;
  hipMalloc(&sink, sizeof(float) * 512 * blocks);
  hipMalloc(&cyc, sizeof(long long) * 8 * blocks);
#define RUN(A, B, na, nb) run<A, B>(na, nb, sink, cyc, blocks)
  RUN(0, 0, "v_mfma_f32_16x16x4_f32", "v_rcp_f32");
  RUN(0, 1, "v_mfma_f32_16x16x4_f32", "v_pk_mul_f32");
  RUN(0, 2, "v_mfma_f32_16x16x4_f32", "v_mul_f32");
  RUN(0, 3, "v_mfma_f32_16x16x4_f32", "v_and_b32");
  RUN(0, 4, "v_mfma_f32_16x16x4_f32", "v_cvt_pk_bf16_f32");
  RUN(0, 5, "v_mfma_f32_16x16x4_f32", "v_pk_fma_f32");
  RUN(0, 6, "v_mfma_f32_16x16x4_f32", "v_add_f32");
  RUN(1, 0, "v_mfma_f32_4x4x1_16b_f32", "v_rcp_f32");
  RUN(1, 1, "v_mfma_f32_4x4x1_16b_f32", "v_pk_mul_f32");
  RUN(2, 0, "v_mfma_f32_16x16x32_bf16", "v_rcp_f32");
  RUN(2, 1, "v_mfma_f32_16x16x32_bf16", "v_pk_mul_f32");
  RUN(2, 3, "v_mfma_f32_16x16x32_bf16", "v_and_b32");
  return 0;
}
