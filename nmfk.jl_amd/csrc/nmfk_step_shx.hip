// Shared-X packed-VALU half-step of libnmfk_hip for the small ranks (gfx950 only; fp32, dense X, no missing data).
//
// The reference's half-step (src/NMFkMultiplicative.jl:67,70) per lane element l and loop step d is
//     p = <a_l, b_d>;   q = x[l,d] / p;   num_l += q * b_d                          (2k FMAs + one reciprocal)
// and the per-rank kernel (step_kernel<KP>, nmfk_step_impl.h) runs it for ONE restart per workgroup: at k <= 8 every
// restart then re-reads its own copy of X from L2 (33.5 MB per restart and iteration at 8192 x 512: 10.6 TB/s of L2
// reads for the ranks 2..8 of the bench sweep, k <= 4 "waits on the loads").  Here a workgroup owns a lane tile for UN
// restarts of the same rank at once: a thread loads its two X entries of a loop step ONCE and applies them to all UN
// restarts, whose lane-factor rows and numerators it keeps in VGPRs (4 * UN * KP registers, UN * KP <= 16) and whose
// loop-factor rows arrive as wave-uniform scalars (UN scalar loads of KP dwords per loop step).  X traffic per restart
// falls by UN; the arithmetic per restart is unchanged and so are the results: the operation order of a restart is
// exactly that of step_kernel<KP> (same packed lanes, same chain of FMAs, same cross-wave order), so the two kernels
// agree bit for bit (tests/test_gpu_parity.py::test_shared_x_kernel_bitwise_equal).
//
// Geometry, sum tables, partial numerators and the fused finish are those of step_body (nmfk_step_impl.h): the grid is
// (lane tile x loop split, group of UN restarts), a stopped restart of a group is still computed (its rows are in the
// wave's registers anyway) but never written.
#include "nmfk_common.h"
#include "../../include/nmfk_hip.h"
#include <algorithm>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SHX_PTR(TYPE, off) ((TYPE *)(arena + (off)))

__device__ __forceinline__ f32x2 shx_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ double shx_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

template <int KP, int UN>
__global__ __launch_bounds__(2 * NMFK_TILE, 4) void shx_step_kernel(char *arena, const float *__restrict__ X,
                                                                   const NmfkRun *__restrict__ runs,
                                                                   const NmfkState *__restrict__ state,
                                                                   const NmfkStepArgs *__restrict__ gp, const int it,
                                                                   const int u0, const int cnt) {
  constexpr int LB = 2;
  constexpr int KE = KP * UN;                              // signals held per lane element
  constexpr int U = KE <= 6 ? 4 : (KE <= 12 ? 2 : 1);      // loop steps per group (two groups of rows live in SGPRs)
  extern __shared__ double lds[];                          // den[64] | red[8 * 64] | cross-wave scratch
  const int grp = blockIdx.y, bx = blockIdx.x;
  const int ub = u0 + grp * UN;
  const int nu = min(UN, cnt - grp * UN);                  // restarts present in this group (the last one may be short)
  const int force = gp->force;
  unsigned act = 0;                                        // restarts still iterating
#pragma unroll
  for (int j = 0; j < UN; ++j)
    if (j < nu && (force || state[ub + j].active)) act |= 1u << j;
  if (act == 0) return;

  const int which = gp->which, ws = gp->wsplit, S = gp->S, L = gp->L, D = gp->D;
  const int64_t ld = gp->ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lpw = ws > 1 ? 64 : NMFK_TILE;
  const int tile = bx / S, s = bx - tile * S;
  if (tile * lpw * LB >= L) return;
  const int lbase = tile * lpw * LB + (ws > 1 ? lane : tid);

  // factors of the UN restarts (absent ones alias the first: their loads are valid, their results are dropped)
  const float *__restrict__ Bp[UN];
  f32x2 ae[UN][KP], acce[UN][KP];
  bool valid[LB];
  int lc[LB];
#pragma unroll
  for (int e = 0; e < LB; ++e) {
    const int l = lbase + e * lpw;
    valid[e] = l < L;
    lc[e] = valid[e] ? l : 0;
  }
#pragma unroll
  for (int j = 0; j < UN; ++j) {
    const NmfkRun *rdp = runs + ub + (j < nu ? j : 0);
    const float *Hcur = SHX_PTR(const float, NMFK_HOFF(*rdp, it));
    const float *Hnew = SHX_PTR(const float, NMFK_HOFF(*rdp, it + 1));
    const float *Wt = SHX_PTR(const float, rdp->oWt);
    const float *A = which == 0 ? Hcur : Wt;
    Bp[j] = which == 0 ? Wt : Hnew;
#pragma unroll
    for (int c = 0; c < KP; ++c) {
#pragma unroll
      for (int e = 0; e < LB; ++e) ae[j][c][e] = valid[e] ? A[c + (int64_t)lc[e] * KP] : 1.0f;
      acce[j][c] = (f32x2)(0.0f);
    }
  }

  int d0 = s * gp->dchunk;
  int d1 = min(D, d0 + gp->dchunk);
  if (ws > 1) {  // the ws waves of the workgroup share the lane elements and split the loop range
    const int q = (d1 - d0 + ws - 1) / ws;
    d0 = min(d0 + wave * q, d1);
    d1 = min(d0 + q, d1);
  }
  d0 = __builtin_amdgcn_readfirstlane(d0);
  d1 = __builtin_amdgcn_readfirstlane(d1);

  // X through buffer loads: resource base = first row of the group (SGPRs), row within the group = soffset, per-lane
  // part = a loop-invariant 32-bit byte offset
  unsigned lbyte[LB];
#pragma unroll
  for (int e = 0; e < LB; ++e) lbyte[e] = (unsigned)lc[e] * 4u;
  const int ldb = (int)(ld * 4);
  const float *__restrict__ xnext = X + (int64_t)d0 * ld;
  int bofs = d0 * KP;  // element offset of the next group's rows in every restart's loop factor

  auto load = [&](float (&bufv)[U][UN][KP], float (&bufx)[U][LB]) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)xnext, 0, -1, 0x00020000);
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
#pragma unroll
      for (int j = 0; j < UN; ++j)
#pragma unroll
        for (int c = 0; c < KP; ++c) bufv[uu][j][c] = Bp[j][bofs + uu * KP + c];
#pragma unroll
      for (int e = 0; e < LB; ++e)
        bufx[uu][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lbyte[e], uu * ldb, 0));
    }
    xnext += U * ld;
    bofs += U * KP;
  };
  // one loop step for all restarts: the X pair is shared, everything else is per restart (same order as step_body)
  auto row = [&](const float (&bv)[UN][KP], const float (&xf)[LB]) __attribute__((always_inline)) {
    const f32x2 x2 = {xf[0], xf[1]};
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      f32x2 p2 = (f32x2)(0.0f);
#pragma unroll
      for (int c = 0; c < KP; ++c) p2 = shx_fma2(ae[j][c], (f32x2)(bv[j][c]), p2);
      const f32x2 r2 = {__builtin_amdgcn_rcpf(p2.x), __builtin_amdgcn_rcpf(p2.y)};
      const f32x2 q2 = x2 * r2;
#pragma unroll
      for (int c = 0; c < KP; ++c) acce[j][c] = shx_fma2((f32x2)(bv[j][c]), q2, acce[j][c]);
    }
  };
  auto compute = [&](const float (&bufv)[U][UN][KP], const float (&bufx)[U][LB]) __attribute__((always_inline)) {
#pragma unroll
    for (int uu = 0; uu < U; ++uu) row(bufv[uu], bufx[uu]);
  };

  int d = d0;
  const int nfull = (d1 - d) / U;
  if (nfull > 0) {
    float v0[U][UN][KP], v1[U][UN][KP];
    float x0[U][LB], x1[U][LB];
    load(v0, x0);
    for (int pairs = (nfull - 1) >> 1; pairs > 0; --pairs) {
      load(v1, x1);
      __builtin_amdgcn_sched_barrier(0);
      compute(v0, x0);
      __builtin_amdgcn_sched_barrier(0);
      load(v0, x0);
      __builtin_amdgcn_sched_barrier(0);
      compute(v1, x1);
      __builtin_amdgcn_sched_barrier(0);
      d += 2 * U;
    }
    if (((nfull - 1) & 1) != 0) {
      load(v1, x1);
      compute(v0, x0);
      compute(v1, x1);
      d += 2 * U;
    } else {
      compute(v0, x0);
      d += U;
    }
  }
  for (; d < d1; ++d) {  // remainder rows
    float bv[UN][KP], xf[LB];
#pragma unroll
    for (int j = 0; j < UN; ++j)
#pragma unroll
      for (int c = 0; c < KP; ++c) bv[j][c] = Bp[j][(int64_t)d * KP + c];
#pragma unroll
    for (int e = 0; e < LB; ++e) xf[e] = X[(int64_t)d * ld + lc[e]];
    row(bv, xf);
  }

  // ws > 1: numerators of waves 1.. are added to wave 0's in wave order (deterministic, same order as step_body)
  float *ldsT = (float *)(lds + 9 * NMFK_MAX_K);
  if (ws > 1) {
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < UN; ++j)
#pragma unroll
        for (int e = 0; e < LB; ++e)
#pragma unroll
          for (int c = 0; c < KP; ++c) ldsT[((((wave - 1) * UN + j) * LB + e) * KP + c) * 64 + lane] = acce[j][c][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll 1
      for (int w = 0; w < ws - 1; ++w) {
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
          for (int e = 0; e < LB; ++e)
#pragma unroll
            for (int c = 0; c < KP; ++c) acce[j][c][e] += ldsT[(((w * UN + j) * LB + e) * KP + c) * 64 + lane];
      }
    }
    __syncthreads();
  }
  const bool owner = (ws == 1) || (wave == 0);

  if (!gp->fused) {  // loop range split over workgroups: partial numerators, reduce_kernel finishes
    if (owner) {
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        if (!((act >> j) & 1u)) continue;
        float *__restrict__ part = SHX_PTR(float, runs[ub + j].opart);
#pragma unroll
        for (int e = 0; e < LB; ++e)
          if (valid[e]) {
#pragma unroll
            for (int c = 0; c < KP; ++c) part[((int64_t)s * L + lc[e]) * KP + c] = acce[j][c][e];
          }
      }
    }
    return;
  }

  // fused finish (Mult:67 / Mult:70, same operation order): A_new = A .* numerator ./ sumB, per-tile sums of A_new
  const int PB = which == 0 ? gp->PW : gp->PH;
  double *den = lds, *red = lds + NMFK_MAX_K;  // den[KE], red[8][KE]
  if (tid < KE) {
    const int j = tid / KP, c = tid - j * KP;
    const NmfkRun *rdp = runs + ub + (j < nu ? j : 0);
    const double *sumB = SHX_PTR(const double, which == 0 ? rdp->osumW : rdp->osumH);
    double sd = 0;
    for (int pp = 0; pp < PB; ++pp) sd += sumB[pp * KP + c];
    den[tid] = sd;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < UN; ++j) {
    const bool on = (act >> j) & 1u;
    const NmfkRun *rdp = runs + ub + (j < nu ? j : 0);
    float *__restrict__ Anew = which == 0 ? SHX_PTR(float, NMFK_HOFF(*rdp, it + 1)) : SHX_PTR(float, rdp->oWt);
    const int k = rdp->k;
#pragma unroll
    for (int c = 0; c < KP; ++c) {
      float vs = 0.0f;
      if (owner) {
#pragma unroll
        for (int e = 0; e < LB; ++e) {
          float v = ae[j][c][e] * acce[j][c][e] / (float)den[j * KP + c];
          if (c >= k || !valid[e]) v = 0.0f;
          if (valid[e] && on) Anew[c + (int64_t)lc[e] * KP] = v;
          vs += v;
        }
      }
      const double v = shx_wave_sum((double)vs);
      if (lane == 0) red[wave * KE + j * KP + c] = v;
    }
  }
  __syncthreads();
  if (tid < KE) {
    const int j = tid / KP, c = tid - j * KP;
    if ((act >> j) & 1u) {
      const NmfkRun *rdp = runs + ub + j;
      double *sumA = SHX_PTR(double, which == 0 ? rdp->osumH : rdp->osumW) + (int64_t)tile * KP;
      sumA[c] = (ws > 1) ? red[tid] : ((red[tid] + red[KE + tid]) + (red[2 * KE + tid] + red[3 * KE + tid]));
    }
  }
}

template <int KP, int UN>
void launch_shx(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int u0, int cnt, hipStream_t s) {
  constexpr int LB = 2;
  const int ws = a.wsplit;
  const int lpw = ws > 1 ? 64 : NMFK_TILE;
  const int ntile = (a.L + lpw * LB - 1) / (lpw * LB);
  const int ngrp = (cnt + UN - 1) / UN;
  const dim3 grid(ntile * a.S, ngrp), blk(ws > 1 ? 64 * ws : NMFK_TILE);
  const size_t scratch = ws > 1 ? (size_t)(ws - 1) * UN * LB * KP * 64 * sizeof(float) : 0;
  const size_t ldsb = sizeof(double) * 9 * NMFK_MAX_K + scratch;
  hipLaunchKernelGGL((shx_step_kernel<KP, UN>), grid, blk, ldsb, s, a.arena, a.X, a.runs, a.state, dargs, a.it, u0, cnt);
}

}  // namespace

// restarts per workgroup for rank k (0: the rank has no shared-X instantiation)
int nmfk_shx_width(int k) {
  switch (k) {
    case 2: return 8;
    case 3: return 4;
    case 4: return 4;
    case 5: return 3;
    case 6: case 7: case 8: return 2;
    default: return 0;
  }
}

void nmfk_launch_step_shx_f32(const NmfkStepArgs &a, const NmfkStepArgs *dargs, int kp, int u0, int cnt, hipStream_t s) {
  switch (kp) {
    case 2: launch_shx<2, 8>(a, dargs, u0, cnt, s); break;
    case 3: launch_shx<3, 4>(a, dargs, u0, cnt, s); break;
    case 4: launch_shx<4, 4>(a, dargs, u0, cnt, s); break;
    case 5: launch_shx<5, 3>(a, dargs, u0, cnt, s); break;
    case 6: launch_shx<6, 2>(a, dargs, u0, cnt, s); break;
    case 7: launch_shx<7, 2>(a, dargs, u0, cnt, s); break;
    case 8: launch_shx<8, 2>(a, dargs, u0, cnt, s); break;
    default: break;
  }
}
