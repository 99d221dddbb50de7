// Does the three-MFMA split product  N[c][l] = sum_d q[d][l] * b[d][c]  (16 loop steps) reproduce fp32?  One wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned pack2(f32x2 v) { bf16x2 r = {(__bf16)v.x, (__bf16)v.y}; return __builtin_bit_cast(unsigned, r); }
__device__ f32x2 unpack2(unsigned w) { return (f32x2){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; }
__device__ void split3_pair(float v0, float v1, unsigned &h, unsigned &m, unsigned &l) {
  f32x2 v = {v0, v1}; h = pack2(v); f32x2 r1 = v - unpack2(h); m = pack2(r1); f32x2 r2 = r1 - unpack2(m); l = pack2(r2);
}
__global__ void k(const float *b, const float *q, float *out) {  // b[d][c], q[d][l]
  const int lane = threadIdx.x, g = lane >> 4, c16 = lane & 15;
  // A operand: signal c16, loop steps 4g..4g+3
  unsigned th[2], tm[2], tl[2];
  for (int pr = 0; pr < 2; ++pr) split3_pair(b[(4 * g + 2 * pr) * 16 + c16], b[(4 * g + 2 * pr + 1) * 16 + c16], th[pr], tm[pr], tl[pr]);
  unsigned h[2], m[2], l[2];
  for (int pr = 0; pr < 2; ++pr) split3_pair(q[(4 * g + 2 * pr) * 16 + c16], q[(4 * g + 2 * pr + 1) * 16 + c16], h[pr], m[pr], l[pr]);
  u32x4 n0 = {th[0], th[1], th[0], th[1]}, n1 = {tm[0], tm[1], tm[0], tm[1]}, n2 = {tl[0], tl[1], th[0], th[1]};
  u32x4 b1 = {h[0], h[1], m[0], m[1]}, b3 = {h[0], h[1], l[0], l[1]};
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, n0), __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, n1), __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, n2), __builtin_bit_cast(bf16x8, b3), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + c16] = acc[r];  // signal 4g + r, lane element c16
}
int main() {
  float hb[256], hq[256], ho[256], *db, *dq, *dout;
  srand(1);
  for (int i = 0; i < 256; ++i) { hb[i] = rand() / (float)RAND_MAX; hq[i] = 0.01f + rand() / (float)RAND_MAX; }
  hipMalloc(&db, 1024); hipMalloc(&dq, 1024); hipMalloc(&dout, 1024);
  hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice); hipMemcpy(dq, hq, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, db, dq, dout);
  hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int c = 0; c < 16; ++c)
    for (int l = 0; l < 16; ++l) {
      double s = 0;
      for (int d = 0; d < 16; ++d) s += (double)hq[d * 16 + l] * hb[d * 16 + c];
      worst = fmax(worst, fabs(ho[c * 16 + l] - s) / s);
    }
  printf("worst relative error %.3g (fp32 eps 6e-8)\n", worst);
  return 0;
}
