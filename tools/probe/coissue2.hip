// Issue-rate probe, part 2 (round 5): (a) one wave per SIMD issues matrix instructions back to back, 1..3 OTHER waves of the SIMD
// independent vector instructions: the aggregate vector rate beside a saturated matrix pipe; (b) ONE stream with n vector instructions
// behind every matrix instruction (what the half-step kernels' loops look like): cycles per matrix instruction.
// Build: hipcc -O3 --offload-arch=gfx950 -w coissue2.hip -o coissue2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TRIPS = 2048, UNROLL = 8;

template <int A>
__device__ __forceinline__ f32x4 mfma(f32x4 acc, float a, float b, bf16x8 ab, bf16x8 bb) {
  if (A == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  if (A == 1) return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0);
}
template <int B>
__device__ __forceinline__ void valu(float &v, f32x2 &p, f32x2 c) {
  if (B == 0) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
  if (B == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(c));
  if (B == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v) : "v"(c[0]));
  if (B == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(c[0]));
  if (B == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(c[0]));
  if (B == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(c));
  if (B == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c[0]));
  if (B == 7) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v) : "v"(c[0]));
  if (B == 8) asm volatile("v_rcp_f32 %0, %0\n\tv_mul_f32 %1, %1, %2" : "+v"(v), "+v"(p[0]) : "v"(c[0]));
  if (B == 9) asm volatile("v_rcp_f32 %0, %0\n\tv_mul_f32 %1, %1, %3\n\tv_mul_f32 %2, %2, %3" : "+v"(v), "+v"(p[0]), "+v"(p[1]) : "v"(c[0]));
  if (B == 10) asm volatile("v_rcp_f32 %0, %0\n\tv_pk_mul_f32 %1, %1, %2" : "+v"(v), "+v"(p) : "v"(c));
}

// NM matrix waves per SIMD (waves 0 .. 4 NM - 1), the others vector waves; N vector instructions per matrix instruction in the
// matrix waves' own stream (0 = pure)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// one stream: per matrix instruction NL memory instructions of kind M (0: ds_read_b128, 1: buffer_load_dwordx4 of a 16 KB window),
// consumed 8 instructions later
template <int A, int M, int NL>
__global__ __launch_bounds__(1024) void probe_mem(const float *src, float seed, float *sink) {
  extern __shared__ char dynlds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  float a = seed, b = seed * 0.5f;
  bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {8, 7, 6, 5, 4, 3, 2, 1};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, -1, 0x00020000);
  for (int i = threadIdx.x; i < 16384; i += 1024) ((float *)dynlds)[i] = seed + i;
  __syncthreads();
  f32x4 ring[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ring[j] = f32x4{seed, 0, 0, 0};
  float keep = 0;
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j & 3] = mfma<A>(acc[j & 3], a, b, ab, bb);
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        const int slot = (j * NL + q) & 7;
        keep += ring[slot][0];  // (one v_add per load: the consumer)
        const int off = ((wave * 4 + ((t + slot) & 3)) * 64 + lane) * 16;
        if (M == 0) ring[slot] = *(const f32x4 *)(dynlds + off);
        else ring[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      }
    }
  }
  float s2 = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + keep;
#pragma unroll
  for (int j = 0; j < 8; ++j) s2 += ring[j][1];
  sink[(size_t)blockIdx.x * 1024 + threadIdx.x] = s2;
}
template <int A, int M, int NL>
void run_mem(const char *na, const float *src, float *sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void *)probe_mem<A, M, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((probe_mem<A, M, NL>), dim3(256), dim3(1024), 65536, 0, src, 1.5f, sink);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe_mem<A, M, NL>), dim3(256), dim3(1024), 65536, 0, src, 1.5f, sink);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float kms = 0;
  (void)hipEventElapsedTime(&kms, e0, e1);
  printf("%-26s x4 waves, %d x %-22s (+ 1 v_add each) in its stream | kernel %.1f us = %.1f ns per matrix instruction of a wave\n", na, NL,
         M == 0 ? "ds_read_b128" : "buffer_load_dwordx4", kms * 1e3, kms * 1e6 / (TRIPS * UNROLL));
}

template <int A, int B, int N>
__global__ __launch_bounds__(1024) void probe(int nm, int nv, float seed, float *sink, long long *cyc) {
  const int wave = threadIdx.x >> 6;
  const bool isA = wave < 4 * nm;
  if (!isA && wave >= 4 * (nm + nv)) return;
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  float a = seed, b = seed * 0.5f;
  bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {8, 7, 6, 5, 4, 3, 2, 1};
  float v[8];
  f32x2 p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = seed + j; p[j] = f32x2{seed + j, seed - j}; }
  const f32x2 c = {1.0001f, 0.9999f};
  const long long t0 = clock64();
  if (isA) {
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) {
        acc[j & 3] = mfma<A>(acc[j & 3], a, b, ab, bb);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < N; ++q) valu<B>(v[(j * N + q) & 7], p[(j * N + q) & 7], c);
      }
    }
  } else {
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) valu<B>(v[j], p[j], c);
    }
  }
  const long long t1 = clock64();
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j] + p[j][0] + p[j][1];
  sink[(size_t)blockIdx.x * 1024 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 16 + wave] = t1 - t0;
}

template <int A, int B, int N>
void run(const char *na, const char *nb, int nm, int nv, float *sink, long long *cyc) {
  const int blocks = 256;
  (void)hipMemset(cyc, 0, sizeof(long long) * blocks * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<A, B, N>), dim3(blocks), dim3(1024), 0, 0, nm, nv, 1.5f, sink, cyc);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<A, B, N>), dim3(blocks), dim3(1024), 0, 0, nm, nv, 1.5f, sink, cyc);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float kms = 0;
  (void)hipEventElapsedTime(&kms, e0, e1);
  std::vector<long long> h(blocks * 16);
  (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 16, hipMemcpyDeviceToHost);
  double sa = 0, sb = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 16; ++w) (w < 4 * nm ? sa : sb) += (double)h[b * 16 + w];
  const double ia = (double)TRIPS * UNROLL * blocks * 4 * nm, ib = (double)TRIPS * UNROLL * blocks * 4 * (nv ? nv : 1);
  // per SIMD: matrix instructions take sa / ia cycles each per wave; with nm waves sharing the pipe the pipe sees one every (sa / ia) / nm
  printf("%-26s x%d waves, %d x %-13s in its stream | %d vector waves of %-13s | matrix wave: %7.2f cycles per matrix instruction (pipe: one per %6.2f)",
         na, nm, N, nb, nv, nb, sa / ia, sa / ia / nm);
  if (nv) printf(" | vector wave: %6.2f cycles per instruction (SIMD: one per %6.2f)", sb / ib, sb / ib / nv);
  printf(" | kernel %.1f us = %.1f ns per matrix instruction of a wave\n", kms * 1e3, kms * 1e6 / (TRIPS * UNROLL));
}

int main() {
  float *sink; long long *cyc;
  (void)hipMalloc(&sink, sizeof(float) * 1024 * 256);
  (void)hipMalloc(&cyc, sizeof(long long) * 16 * 256);
  const char *F = "v_mfma_f32_16x16x4_f32", *Q = "v_mfma_f32_4x4x1_f32", *H = "v_mfma_f32_16x16x32_bf16";
  run<0, 0, 0>(F, "-", 2, 0, sink, cyc);
  run<0, 0, 0>(F, "-", 4, 0, sink, cyc);
  run<2, 0, 0>(H, "-", 1, 0, sink, cyc);
  run<2, 0, 0>(H, "-", 4, 0, sink, cyc);
  run<1, 0, 0>(Q, "-", 1, 0, sink, cyc);
  run<1, 0, 0>(Q, "-", 4, 0, sink, cyc);
  // one stream: n vector instructions behind every matrix instruction, 4 such waves per SIMD
#define ROW(A, NA) \
  run<A, 0, 1>(NA, "v_rcp_f32", 4, 0, sink, cyc); run<A, 0, 2>(NA, "v_rcp_f32", 4, 0, sink, cyc); run<A, 0, 4>(NA, "v_rcp_f32", 4, 0, sink, cyc); \
  run<A, 1, 1>(NA, "v_pk_mul_f32", 4, 0, sink, cyc); run<A, 1, 2>(NA, "v_pk_mul_f32", 4, 0, sink, cyc); \
  run<A, 3, 1>(NA, "v_mul_f32", 4, 0, sink, cyc); run<A, 3, 2>(NA, "v_mul_f32", 4, 0, sink, cyc); run<A, 3, 4>(NA, "v_mul_f32", 4, 0, sink, cyc); \
  run<A, 7, 2>(NA, "v_mul_f32_e64", 4, 0, sink, cyc); \
  run<A, 2, 1>(NA, "v_and_b32", 4, 0, sink, cyc); run<A, 2, 2>(NA, "v_and_b32", 4, 0, sink, cyc); \
  run<A, 4, 2>(NA, "v_add_f32", 4, 0, sink, cyc); run<A, 5, 1>(NA, "v_pk_add_f32", 4, 0, sink, cyc); run<A, 6, 2>(NA, "v_fma_f32", 4, 0, sink, cyc);
  ROW(2, H)
  ROW(0, F)
  ROW(1, Q)
  // mixed groups
  run<2, 8, 1>(H, "rcp + mul", 4, 0, sink, cyc);
  run<2, 9, 1>(H, "rcp + 2 mul", 4, 0, sink, cyc);
  run<2, 10, 1>(H, "rcp + pk_mul", 4, 0, sink, cyc);
  run<2, 8, 2>(H, "rcp + mul", 4, 0, sink, cyc);
  run<0, 8, 1>(F, "rcp + mul", 4, 0, sink, cyc);
  run<0, 10, 1>(F, "rcp + pk_mul", 4, 0, sink, cyc);
  float *src;
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  run_mem<2, 0, 1>(H, src, sink); run_mem<2, 0, 2>(H, src, sink); run_mem<2, 1, 1>(H, src, sink); run_mem<2, 1, 2>(H, src, sink);
  run_mem<0, 0, 1>(F, src, sink); run_mem<0, 0, 2>(F, src, sink); run_mem<0, 1, 1>(F, src, sink); run_mem<0, 1, 2>(F, src, sink);
  return 0;
}
