// Known hazard: candidate instruction forms of the packed-VALU kernels, each in a loop of its own, run
// repeatedly while ANOTHER process keeps the bf16 matrix pipe busy (tools/hazard/burner 0).  Every launch must reproduce the
// first launch's output bit for bit.
//   0 v_pk_fma_f32 with three VGPR-pair sources      1 v_pk_fma_f32 with one SGPR-pair source
//   2 v_fma_f32 (three VGPR sources)                 3 v_pk_mul_f32 + v_pk_add_f32 (VGPR pairs)
//   4 v_pk_fma_f32, op_sel broadcast of one VGPR     5 v_writelane_b32 / v_readlane_b32 round trips
//   6 v_rcp_f32 + v_pk_mul_f32                       7 v_pk_fma_f32 (three VGPR pairs) with 160 live VGPRs
// hipcc --offload-arch=gfx950 -O2 tools/hazard/pk_victim.hip -o tools/hazard/pk_victim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V, int NR>
__global__ __launch_bounds__(256) void victim(float *out, const float *in, int iters) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  f32x2 a[NR], b[NR], acc[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    a[j] = (f32x2){in[(gid * 7 + j) & 0xfffff], in[(gid * 11 + j + 3) & 0xfffff]};
    b[j] = (f32x2){in[(j * 13 + 1) & 0xfffff], in[(j * 17 + 2) & 0xfffff]};  // wave-uniform
    acc[j] = (f32x2){0.f, 0.f};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (V == 0 || V == 7) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      } else if (V == 1) {
        f32x2 bs;
        bs[0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b[j][0])));
        bs[1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b[j][1])));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j]), "s"(bs));
      } else if (V == 2) {
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][0]) : "v"(a[j][0]), "v"(b[j][0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][1]) : "v"(a[j][1]), "v"(b[j][1]));
      } else if (V == 3) {
        f32x2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[j]), "v"(b[j]));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(t));
      } else if (V == 4) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      } else if (V == 5) {
        int s = __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[j][0]), (it + j) & 63);
        int w = __builtin_bit_cast(int, acc[j][1]);
        asm volatile("v_writelane_b32 %0, %1, 37" : "+v"(w) : "s"(s));
        acc[j][1] = __builtin_bit_cast(float, w & 0x3fffffff);
        acc[j][0] += a[j][0] * b[j][0];
      } else {
        const float r = __builtin_amdgcn_rcpf(a[j][0] + 1.5f + acc[j][0] * 1e-6f);
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[j]) : "v"(a[j]), "v"((f32x2){r, r}));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < NR; ++j) s += acc[j][0] + acc[j][1] * 0.5f;
  out[gid] = s;
}

// N back-to-back v_rcp_f32 (the transcendental pipe) on distinct registers, consumed by packed multiplies right behind
template <int N>
__global__ __launch_bounds__(256) void victim_rcp(float *out, const float *in, int iters) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  float x[8], r[8];
  f32x2 acc[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = 1.0f + 100.0f * in[(gid * 7 + j * 977) & 0xfffff];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x2){0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (N == 2) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        asm volatile("v_rcp_f32 %0, %2\n\tv_rcp_f32 %1, %3" : "=&v"(r[j]), "=&v"(r[j + 1]) : "v"(x[j]), "v"(x[j + 1]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j / 2]) : "v"((f32x2){r[j], r[j + 1]}), "v"((f32x2){x[j], x[j + 1]}));
      }
    } else if (N == 4) {
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        asm volatile("v_rcp_f32 %0, %4\n\tv_rcp_f32 %1, %5\n\tv_rcp_f32 %2, %6\n\tv_rcp_f32 %3, %7"
                     : "=&v"(r[j]), "=&v"(r[j + 1]), "=&v"(r[j + 2]), "=&v"(r[j + 3])
                     : "v"(x[j]), "v"(x[j + 1]), "v"(x[j + 2]), "v"(x[j + 3]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j / 2]) : "v"((f32x2){r[j], r[j + 1]}), "v"((f32x2){x[j], x[j + 1]}));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j / 2 + 1]) : "v"((f32x2){r[j + 2], r[j + 3]}), "v"((f32x2){x[j + 2], x[j + 3]}));
      }
    } else {
      asm volatile(
          "v_rcp_f32 %0, %8\n\tv_rcp_f32 %1, %9\n\tv_rcp_f32 %2, %10\n\tv_rcp_f32 %3, %11\n\t"
          "v_rcp_f32 %4, %12\n\tv_rcp_f32 %5, %13\n\tv_rcp_f32 %6, %14\n\tv_rcp_f32 %7, %15"
          : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
          : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
#pragma unroll
      for (int j = 0; j < 8; j += 2)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j / 2]) : "v"((f32x2){r[j], r[j + 1]}), "v"((f32x2){x[j], x[j + 1]}));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += 0.001f * (float)((it + j) & 7);
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] * 0.5f;
  out[gid] = s;
}

// W = 4: two global_load_dwordx4 in flight (wave-uniform address, like the loop-factor rows of the merged kernel) while
// packed FMAs on OTHER registers execute; W = 1: eight global_load_dword instead (control)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int W>
__global__ __launch_bounds__(256) void victim_ld(float *out, const float *in, int iters) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  f32x2 a[8], b[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = (f32x2){in[(gid * 7 + j) & 0xfffff], in[(gid * 11 + j + 3) & 0xfffff]};
    b[j] = (f32x2){in[(j * 13 + 1) & 0xfffff], in[(j * 17 + 2) & 0xfffff]};
    acc[j] = (f32x2){0.f, 0.f};
  }
  unsigned vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  float fold = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float *p = in + (((size_t)blockIdx.x * 131 + (size_t)it * 7) & 0xffff0);  // wave-uniform
    if (W == 4) {
      u32x4 v0, v1;
      asm volatile("global_load_dwordx4 %0, %2, %3 offset:4\n\tglobal_load_dwordx4 %1, %2, %3 offset:20"
                   : "=&v"(v0), "=&v"(v1) : "v"(vzero), "s"(p) : "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
      fold += __builtin_bit_cast(float, v0[1]) + __builtin_bit_cast(float, v1[2]);
    } else {
      float x[8];
      const float *q = p + (threadIdx.x & 63);
      asm volatile(
          "global_load_dword %0, %8, off\n\tglobal_load_dword %1, %8, off offset:256\n\tglobal_load_dword %2, %8, off offset:512\n\t"
          "global_load_dword %3, %8, off offset:768\n\tglobal_load_dword %4, %8, off offset:1024\n\tglobal_load_dword %5, %8, off offset:1280\n\t"
          "global_load_dword %6, %8, off offset:1536\n\tglobal_load_dword %7, %8, off offset:1792"
          : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]) : "v"(q) : "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"memory");
      fold += x[0] + x[3] + x[7];
    }
  }
  float s = fold;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] * 0.5f;
  out[gid] = s;
}

// The failing loop of the kernel, reduced: two wave-uniform global_load_dwordx4 (SGPR base + zero VGPR offset) and eight
// per-lane dword loads in flight, partial waits, packed FMAs with op_sel broadcasts on two "lane elements" (low / high
// half of the pairs), v_rcp_f32.  F bits: 1 the uniform values are the FMA operands (else: registers), 2 no uniform
// loads, 4 no per-lane loads, 8 plain packed FMAs (no op_sel), 16 no v_rcp_f32, 32 every wait is vmcnt(0)
template <int F>
__global__ __launch_bounds__(256) void victim_use(float *out, const float *in, int iters) {
  constexpr bool USE = F & 1, NOUNI = F & 2, NOLANE = F & 4, PLAIN = F & 8, NORCP = F & 16, FULLWAIT = F & 32;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  f32x2 a0 = {in[(gid * 7) & 0xfffff], in[(gid * 11 + 3) & 0xfffff]}, a1 = {in[(gid * 5 + 9) & 0xfffff], in[(gid * 3 + 1) & 0xfffff]};
  f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  unsigned vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const u32x4 r0 = *(const u32x4 *)(in + 64), r1 = *(const u32x4 *)(in + 68);
  const f32x2 xr = {in[(gid * 13) & 0xfffff], in[(gid * 17) & 0xfffff]};
#define PKB(v) (f32x2){__builtin_bit_cast(float, (v)[0]), __builtin_bit_cast(float, (v)[1])}
  for (int it = 0; it < iters; ++it) {
    const float *p = in + (((size_t)blockIdx.x * 131 + (size_t)it * 8) & 0xffff0);  // wave-uniform rows
    const float *q = in + (((size_t)blockIdx.x * 977 + (size_t)it * 64) & 0x7ff00) + lane;  // + 3328 B stays inside the 4 MB
    u32x4 v0 = r0, v1 = r1;
    f32x2 x0 = xr, x1 = xr, x2 = xr, x3 = xr;
    if (!NOUNI)
      asm volatile("global_load_dwordx4 %0, %2, %3 offset:4\n\tglobal_load_dwordx4 %1, %2, %3 offset:20"
                   : "=&v"(v0), "=&v"(v1) : "v"(vzero), "s"(p) : "memory");
    if (!NOLANE) {
      asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:1024\n\t"
                   "global_load_dword %2, %4, off offset:2048\n\tglobal_load_dword %3, %4, off offset:3072"
                   : "=&v"(x0[0]), "=&v"(x0[1]), "=&v"(x1[0]), "=&v"(x1[1]) : "v"(q) : "memory");
      asm volatile("global_load_dword %0, %4, off offset:256\n\tglobal_load_dword %1, %4, off offset:1280\n\t"
                   "global_load_dword %2, %4, off offset:2304\n\tglobal_load_dword %3, %4, off offset:3328"
                   : "=&v"(x2[0]), "=&v"(x2[1]), "=&v"(x3[0]), "=&v"(x3[1]) : "v"(q) : "memory");
    }
    if (FULLWAIT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"memory");
    else if (!NOUNI) asm volatile("s_waitcnt vmcnt(9)" : "+v"(v0)::"memory");
    const u32x4 b0 = USE ? v0 : r0;
    f32x2 p0, p1;
    if (PLAIN) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p0) : "v"(a0), "v"(PKB(b0)), "v"(xr));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(a1), "v"(PKB(b0)));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(p0) : "v"(a0), "v"(PKB(b0)));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p0) : "v"(a1), "v"(PKB(b0)));
    }
    if (!FULLWAIT && !NOUNI) asm volatile("s_waitcnt vmcnt(8)" : "+v"(v1)::"memory");
    const u32x4 b1 = USE ? v1 : r1;
    if (PLAIN) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p1) : "v"(a0), "v"(PKB(b1)), "v"(xr));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(a1), "v"(PKB(b1)));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(a0), "v"(PKB(b1)));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p1) : "v"(a1), "v"(PKB(b1)));
    }
    f32x2 rc0, rc1;
    if (NORCP) {
      rc0 = p0 * 0.25f; rc1 = p1 * 0.25f;
    } else {
      rc0 = (f32x2){__builtin_amdgcn_rcpf(p0[0] + 1.0f), __builtin_amdgcn_rcpf(p0[1] + 1.0f)};
      rc1 = (f32x2){__builtin_amdgcn_rcpf(p1[0] + 1.0f), __builtin_amdgcn_rcpf(p1[1] + 1.0f)};
    }
    if (!FULLWAIT && !NOLANE) asm volatile("s_waitcnt vmcnt(6)" : "+v"(x0)::"memory");
    f32x2 q0 = x0 * rc0;
    if (PLAIN) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(PKB(b0)), "v"(q0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(PKB(b0)), "v"(q0));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc0) : "v"(PKB(b0)), "v"(q0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc1) : "v"(PKB(b0)), "v"(q0));
    }
    if (!FULLWAIT && !NOLANE) asm volatile("s_waitcnt vmcnt(4)" : "+v"(x1)::"memory");
    f32x2 q1 = x1 * rc1;
    if (PLAIN) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(PKB(b1)), "v"(q1));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(PKB(b1)), "v"(q1));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc0) : "v"(PKB(b1)), "v"(q1));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc1) : "v"(PKB(b1)), "v"(q1));
    }
    if (!FULLWAIT && !NOLANE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x2), "+v"(x3)::"memory");
    acc0 += x2 * 1e-3f;
    acc1 += x3 * 1e-3f;
  }
#undef PKB
  out[gid] = acc0[0] + 0.5f * acc0[1] + 0.25f * acc1[0] + 0.125f * acc1[1];
}

// Register-only loops of ONE packed instruction form each (G): which operand selects matter?
//   0 v_pk_fma_f32 op_sel_hi:[1,0,1] (VGPR)      1 v_pk_fma_f32 op_sel:[0,1,0] on a VGPR src1      2 v_pk_fma_f32 op_sel:[1,0,0] on a VGPR src0
//   3 v_pk_fma_f32 op_sel:[0,1,0] on an SGPR src1  4 v_pk_mul_f32 op_sel:[0,1]  (VGPR)              5 v_pk_add_f32 op_sel:[0,1] (VGPR)
//   6 v_pk_fma_f32 op_sel:[0,0,1] on the VGPR accumulator          7 v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (swap the halves of src1)
//   8 v_pk_add_f32 op_sel:[0,1] on a VGPR src1                     9 v_pk_mov_b32 op_sel:[1,0] op_sel_hi:[0,1]
template <int G>
__global__ __launch_bounds__(256) void victim_sel(float *out, const float *in, int iters) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  f32x2 a[4], b[4], acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = (f32x2){in[(gid * 7 + j) & 0xfffff], in[(gid * 11 + j + 3) & 0xfffff]};
    b[j] = (f32x2){in[(gid * 13 + j * 5 + 1) & 0xfffff], in[(gid * 17 + j * 3 + 2) & 0xfffff]};
    acc[j] = (f32x2){0.f, 0.f};
  }
  f32x2 bs;
  bs[0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b[0][0])));
  bs[1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b[0][1])));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (G == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      if (G == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      if (G == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      if (G == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[j]) : "v"(a[j]), "s"(bs));
      if (G == 4) {
        f32x2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(a[j]), "v"(b[j]));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(t));
      }
      if (G == 5) {
        f32x2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[j]), "v"(b[j]));
        asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(acc[j]) : "v"(t));
      }
      if (G == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      if (G == 8) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(acc[j]) : "v"(a[j]), "v"(b[j]));
      if (G == 9) {
        f32x2 t;
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a[j]), "v"(b[j]));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(t));
      }
      if (G == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(a[j]), "v"(b[j]));
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] * 0.5f;
  out[gid] = s;
}

template <int V, int NR>
static void run(const char *name, float *d_out, const float *d_in, int nwg, int iters, int reps, int rcpn = 0) {
  const size_t n = (size_t)nwg * 256;
  std::vector<float> ref(n), cur(n);
  int bad = 0;
  long firstbad = -1;
  int badlanes = 0;
  for (int r = 0; r < reps; ++r) {
    if (rcpn == 2)
      hipLaunchKernelGGL((victim_rcp<2>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    else if (rcpn == 4)
      hipLaunchKernelGGL((victim_rcp<4>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    else if (rcpn == 8)
      hipLaunchKernelGGL((victim_rcp<8>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    else if (rcpn == 104)
      hipLaunchKernelGGL((victim_ld<4>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    else if (rcpn == 101)
      hipLaunchKernelGGL((victim_ld<1>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
#define SELCASE(G) else if (rcpn == 300 + G) hipLaunchKernelGGL((victim_sel<G>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    SELCASE(0) SELCASE(1) SELCASE(2) SELCASE(3) SELCASE(4) SELCASE(5) SELCASE(6) SELCASE(7) SELCASE(8) SELCASE(9)
#undef SELCASE
#define USECASE(F) else if (rcpn == 200 + F) hipLaunchKernelGGL((victim_use<F>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    USECASE(0) USECASE(1) USECASE(2) USECASE(4) USECASE(6) USECASE(8) USECASE(16) USECASE(24) USECASE(32) USECASE(10) USECASE(12) USECASE(28) USECASE(30)
#undef USECASE
    else
      hipLaunchKernelGGL((victim<V, NR>), dim3(nwg), dim3(256), 0, 0, d_out, d_in, iters);
    (void)hipMemcpy(r == 0 ? ref.data() : cur.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    if (r > 0 && memcmp(ref.data(), cur.data(), n * sizeof(float)) != 0) {
      if (bad == 0)
        for (size_t i = 0; i < n; ++i)
          if (memcmp(&ref[i], &cur[i], 4) != 0) {
            if (firstbad < 0) firstbad = (long)i;
            ++badlanes;
          }
      ++bad;
    }
  }
  printf("variant %d (%s): %d of %d launches differ from the first", V, name, bad, reps - 1);
  if (bad) printf("; first differing launch: %d elements, first at thread %ld (lane %ld of its wave)", badlanes, firstbad, firstbad & 63);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 300;
  const int nwg = 2048, iters = 4000;
  float *d_in, *d_out;
  (void)hipMalloc((void **)&d_in, sizeof(float) << 20);
  (void)hipMalloc((void **)&d_out, sizeof(float) * nwg * 256);
  std::vector<float> h(1 << 20);
  unsigned s = 12345;
  for (auto &v : h) {
    s = s * 1664525u + 1013904223u;
    v = 0.001f + (s >> 8) * (1.0f / 16777216.0f) * 0.01f;
  }
  (void)hipMemcpy(d_in, h.data(), sizeof(float) << 20, hipMemcpyHostToDevice);
  if (argc <= 2) {
  run<0, 8>("v_pk_fma_f32 vvv", d_out, d_in, nwg, iters, reps);
  run<1, 8>("v_pk_fma_f32 vsv", d_out, d_in, nwg, iters, reps);
  run<2, 8>("v_fma_f32 vvv", d_out, d_in, nwg, iters, reps);
  run<3, 8>("v_pk_mul_f32 + v_pk_add_f32", d_out, d_in, nwg, iters, reps);
  run<4, 8>("v_pk_fma_f32 op_sel broadcast", d_out, d_in, nwg, iters, reps);
  run<5, 8>("v_readlane/v_writelane", d_out, d_in, nwg, iters / 4, reps);
  run<6, 8>("v_rcp_f32 + v_pk_mul_f32", d_out, d_in, nwg, iters, reps);
  run<7, 26>("v_pk_fma_f32 vvv, 160 VGPRs", d_out, d_in, nwg, iters / 3, reps);
  run<8, 1>("2 back-to-back v_rcp_f32 + packed consumers", d_out, d_in, nwg, iters, reps, 2);
  run<9, 1>("4 back-to-back v_rcp_f32 + packed consumers", d_out, d_in, nwg, iters, reps, 4);
  run<10, 1>("8 back-to-back v_rcp_f32 + packed consumers", d_out, d_in, nwg, iters, reps, 8);
  run<11, 1>("packed FMAs under two uniform global_load_dwordx4", d_out, d_in, nwg, iters / 2, reps, 104);
  run<12, 1>("packed FMAs under eight global_load_dword", d_out, d_in, nwg, iters / 2, reps, 101);
  }
  if (argc > 2) {  // only the reduced loop and its ablations
    printf("reduced loop of the failing kernel (F bits: 1 loaded operands, 2 no uniform loads, 4 no per-lane loads, 8 no op_sel, 16 no v_rcp_f32, 32 vmcnt(0) waits)\n");
  }
  run<13, 1>("reduced loop, F=1: uniform vector loads USED by the packed FMAs", d_out, d_in, nwg, iters / 4, reps, 201);
  run<14, 1>("reduced loop, F=0: the same loads, operands from registers", d_out, d_in, nwg, iters / 4, reps, 200);
  run<15, 1>("F=2: no uniform loads", d_out, d_in, nwg, iters / 4, reps, 202);
  run<16, 1>("F=4: no per-lane loads", d_out, d_in, nwg, iters / 4, reps, 204);
  run<17, 1>("F=6: no loads at all", d_out, d_in, nwg, iters / 4, reps, 206);
  run<18, 1>("F=8: plain packed FMAs (no op_sel)", d_out, d_in, nwg, iters / 4, reps, 208);
  run<19, 1>("F=16: no v_rcp_f32", d_out, d_in, nwg, iters / 4, reps, 216);
  run<20, 1>("F=24: no op_sel, no v_rcp_f32", d_out, d_in, nwg, iters / 4, reps, 224);
  run<21, 1>("F=32: every wait is vmcnt(0)", d_out, d_in, nwg, iters / 4, reps, 232);
  run<22, 1>("F=10: no uniform loads, no op_sel", d_out, d_in, nwg, iters / 4, reps, 210);
  run<23, 1>("F=12: no per-lane loads, no op_sel", d_out, d_in, nwg, iters / 4, reps, 212);
  run<24, 1>("F=28: no per-lane loads, no op_sel, no v_rcp_f32", d_out, d_in, nwg, iters / 4, reps, 228);
  run<25, 1>("F=30: no loads, no op_sel, no v_rcp_f32", d_out, d_in, nwg, iters / 4, reps, 230);
  printf("register-only loops of one packed instruction form\n");
  run<30, 1>("G=0: v_pk_fma_f32 op_sel_hi:[1,0,1], VGPR sources", d_out, d_in, nwg, iters, reps, 300);
  run<31, 1>("G=1: v_pk_fma_f32 op_sel:[0,1,0] on a VGPR src1", d_out, d_in, nwg, iters, reps, 301);
  run<32, 1>("G=2: v_pk_fma_f32 op_sel:[1,0,0] on a VGPR src0", d_out, d_in, nwg, iters, reps, 302);
  run<33, 1>("G=3: v_pk_fma_f32 op_sel:[0,1,0] on an SGPR src1", d_out, d_in, nwg, iters, reps, 303);
  run<34, 1>("G=4: v_pk_mul_f32 op_sel:[0,1] on a VGPR src1", d_out, d_in, nwg, iters, reps, 304);
  run<35, 1>("G=5: v_pk_add_f32 op_sel:[1,0] on a VGPR src0", d_out, d_in, nwg, iters, reps, 305);
  run<36, 1>("G=6: v_pk_fma_f32 op_sel:[0,0,1] on the VGPR accumulator", d_out, d_in, nwg, iters, reps, 306);
  run<37, 1>("G=7: v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (halves of src1 swapped)", d_out, d_in, nwg, iters, reps, 307);
  run<38, 1>("G=8: v_pk_add_f32 op_sel:[0,1] on a VGPR src1", d_out, d_in, nwg, iters, reps, 308);
  run<39, 1>("G=9: v_pk_mov_b32 op_sel:[1,0] op_sel_hi:[0,1]", d_out, d_in, nwg, iters, reps, 309);
  return 0;
}
