// Do all 64 lanes of a wave get the SAME data from a wave-uniform, 4-byte-aligned (not 16-byte-aligned)
// global_load_dwordx4 (SGPR base + zero VGPR offset), while other kernels keep the memory pipeline busy?
// The memory read is constant.  Counts lanes whose 16 bytes differ from lane 0's, and which lanes.
// usage: uniform_vload [seconds]   (run tools/hazard/dbg_twoproc.sh's burner, or any other GPU load, beside it)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void victim(const float *__restrict__ tab, const float *__restrict__ big, int rows, int iters,
                                              unsigned long long *bad, unsigned *lanehist, float *sink) {
  const int lane = threadIdx.x & 63;
  unsigned vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  float acc = 0.f;
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const int r = (blockIdx.x * 131 + it * 7) % rows;
    const float *p = tab + (size_t)r * 2;  // rows of 2 floats: the 16-byte load below starts 4 bytes into a row
    // like the failing kernel: the uniform 16-byte load FIRST, eight per-lane dword loads behind it, then a wait that
    // leaves the eight outstanding -- is the first load's data complete in every lane?
    u32x4 v = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    float x0, x1, x2, x3, x4, x5, x6, x7;
    const float *q = big + ((size_t)(blockIdx.x * 977 + it * 131) * 64 % (size_t)((1 << 24) - 8 * 4096)) + lane;
    asm volatile(
        "global_load_dwordx4 %0, %9, %10 offset:4\n\t"
        "global_load_dword %1, %11, off\n\t"
        "global_load_dword %2, %11, off offset:1024\n\t"
        "global_load_dword %3, %11, off offset:2048\n\t"
        "global_load_dword %4, %11, off offset:3072\n\t"
        "global_load_dword %5, %12, off\n\t"
        "global_load_dword %6, %12, off offset:1024\n\t"
        "global_load_dword %7, %12, off offset:2048\n\t"
        "global_load_dword %8, %12, off offset:3072\n\t"
        "s_waitcnt vmcnt(8)"
        : "+&v"(v), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
        : "v"(vzero), "s"(p), "v"(q), "v"(q + 4096)
        : "memory");
    u32x4 vnow = v;  // what the registers hold right behind the wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    const u32x4 want = {__float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(p[3]), __float_as_uint(p[4])};
    bool diff = false;
    for (int j = 0; j < 4; ++j) diff |= vnow[j] != want[j];
    if (diff) {
      ++nbad;
      atomicAdd(&lanehist[lane], 1u);
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 20.0;
  const int rows = 1 << 16;
  float *tab, *big, *sink; unsigned long long *bad; unsigned *hist;
  hipMalloc(&tab, (rows * 2 + 16) * 4); hipMalloc(&big, (size_t)(1 << 24) * 4); hipMalloc(&sink, 4096 * 256 * 4);
  hipMalloc(&bad, 8); hipMalloc(&hist, 64 * 4);
  float *h = (float *)malloc((rows * 2 + 16) * 4);
  for (int i = 0; i < rows * 2 + 16; ++i) h[i] = (float)rand() / RAND_MAX;
  hipMemcpy(tab, h, (rows * 2 + 16) * 4, hipMemcpyHostToDevice);
  hipMemset(big, 0, (size_t)(1 << 24) * 4); hipMemset(bad, 0, 8); hipMemset(hist, 0, 256);
  long long launches = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(victim, dim3(4096), dim3(256), 0, 0, tab, big, rows, 200, bad, hist, sink);
    hipDeviceSynchronize();
    launches += 20;
  }
  unsigned long long nb; unsigned hh[64];
  hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 256, hipMemcpyDeviceToHost);
  printf("launches %lld, uniform loads %.3g, lanes that disagreed with lane 0: %llu\n", launches, (double)launches * 4096 * 4 * 200, nb);
  if (nb) { printf("per lane:"); for (int l = 0; l < 64; ++l) printf(" %u", hh[l]); printf("\n"); }
  return 0;
}
