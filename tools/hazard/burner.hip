// Synthetic neighbours for the "Known hazard" of DESIGN.md: which hardware unit does a co-resident kernel have to keep
// busy for the fp32 merged packed-VALU kernel (checker process) to return different bits?
//   burner <mode> <seconds> [vgpr-form is a compile flag: build twice]
//   0 bf16 MFMA 16x16x32 in registers      1 fp32 MFMA 16x16x4 in registers      2 LDS b128 writes + reads
//   3 global loads (L2/HBM streaming)      4 VALU: v_rcp_f32, v_cvt_pk_bf16_f32, v_pk_mul_f32      5 global stores
//   6 bf16 MFMA 16x16x16 (64-bit operands)  7 f16 MFMA 16x16x32      8 bf16 MFMA 32x32x16      9 fp8 MFMA 16x16x32 (64-bit operands)
// hipcc --offload-arch=gfx950 -O2 tools/hazard/burner.hip -o tools/hazard/burner
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(float *out, const u32x4 *in, size_t nin, int iters) {
  __shared__ u32x4 lds[2048];  // 32 KB
  const int tid = threadIdx.x;
  const size_t gid = (size_t)blockIdx.x * 256 + tid;
  u32x4 v = in[gid % nin];
  f32x4 acc = {0, 0, 0, 0};
  if (MODE == 2) {
    for (int i = tid; i < 2048; i += 256) lds[i] = v;
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, v[j & 3]), __builtin_bit_cast(float, v[(j + 1) & 3]), acc, 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x4 r = lds[(tid * 5 + j * 259 + it) & 2047];
        v[0] ^= r[1];
        lds[(tid + 256 * j) & 2047] = v;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x4 r = in[(gid + (size_t)(it * 8 + j) * 262144) % nin];
        v[0] ^= r[1];
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = __builtin_bit_cast(float, v[0]) + 1.5f, b = __builtin_bit_cast(float, v[1]) + 2.5f;
        a = __builtin_amdgcn_rcpf(a);
        b = __builtin_amdgcn_rcpf(b);
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 c = __builtin_convertvector((f32x2){a, b}, bf2);
        v[2] ^= __builtin_bit_cast(unsigned, c);
        f32x2 m = (f32x2){a, b} * (f32x2){b, a};
        v[0] = __builtin_bit_cast(unsigned, m[0]) & 0x3fffffff;
        v[1] = __builtin_bit_cast(unsigned, m[1]) & 0x3fffffff;
      }
    } else if (MODE == 6) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const s16x4 o = __builtin_bit_cast(s16x4, (u32x2){v[0], v[1]});
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(o, o, acc, 0, 0, 0);
    } else if (MODE == 7) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, v), __builtin_bit_cast(f16x8, v), acc, 0, 0, 0);
    } else if (MODE == 8) {
      f32x16 a16;
#pragma unroll
      for (int j = 0; j < 16; ++j) a16[j] = acc[j & 3];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        a16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, v), a16, 0, 0, 0);
      acc = (f32x4){a16[0], a16[5], a16[10], a16[15]};
    } else if (MODE == 9) {
      const long o = (long)v[0] | ((long)v[1] << 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(o, o, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) ((u32x4 *)in)[nin + (gid + (size_t)(it * 8 + j) * 262144) % nin] = v;
    }
  }
  out[gid] = acc[0] + acc[1] + acc[2] + acc[3] + __builtin_bit_cast(float, v[0] ^ v[1] ^ v[2] ^ v[3]);
}

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 30;
  const int nwg = argc > 3 ? atoi(argv[3]) : 1024;
  const size_t nin = (size_t)4 << 20;  // 64 MB of 16-byte items (+ as much again behind it for the store mode)
  u32x4 *in;
  float *out;
  hipMalloc((void **)&in, 2 * nin * sizeof(u32x4));
  hipMalloc((void **)&out, (size_t)nwg * 256 * sizeof(float));
  hipMemset(in, 0x3c, 2 * nin * sizeof(u32x4));
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int r = 0; r < 20; ++r) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(burn<0>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 2000); break;
        case 1: hipLaunchKernelGGL(burn<1>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 1000); break;
        case 2: hipLaunchKernelGGL(burn<2>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 1000); break;
        case 3: hipLaunchKernelGGL(burn<3>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 300); break;
        case 4: hipLaunchKernelGGL(burn<4>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 1000); break;
        case 6: hipLaunchKernelGGL(burn<6>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 2000); break;
        case 7: hipLaunchKernelGGL(burn<7>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 2000); break;
        case 8: hipLaunchKernelGGL(burn<8>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 1000); break;
        case 9: hipLaunchKernelGGL(burn<9>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 2000); break;
        default: hipLaunchKernelGGL(burn<5>, dim3(nwg), dim3(256), 0, 0, out, in, nin, 300); break;
      }
      ++launches;
    }
    hipDeviceSynchronize();
  }
  printf("burner mode %d: %ld launches in %.1f s\n", mode, launches, secs);
  return 0;
}
