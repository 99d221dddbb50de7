// Data preparation and robustness kernels of libnmfk_hip (gfx950, wave64):
//   preprocess        NMFpreprocessing!                     src/NMFkMultiplicative.jl:3-22
//   cluster           clustersolutions(factors, false)      src/NMFkCluster.jl:425-517
//   pairdist/silhouette  finalize(Wa, Ha, idx, false)       src/NMFkFinalize.jl:36-66
//   cluster_stats     cluster means / variances             src/NMFkFinalize.jl:64-77
// All arithmetic is fp32: the reference stores restart results as Matrix{Float32} for Float32 X
// (src/NMFkExecute.jl:529-531) and clusters in that type (Clus:463).
#include "nmfk_common.h"
#include "../../include/nmfk_hip.h"
#include "nmfk_rng.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;  // every lane holds the total
}

// ---------------------------------------------------------------------------------------------------
// X -> Xc (column-major, element (i,j) at i + j*n) and Xr (row-major, j + i*m); X <= 0 -> lambda; NaN kept
// as the "missing" marker.  counts: [0] negatives (=> error, Mult:4-7), [1] NaNs, [2] entries <= 0.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const float *__restrict__ Xin, int64_t ldx, int64_t n, int64_t m,
                                                         float lambda, float *__restrict__ Xc, float *__restrict__ Xr,
                                                         unsigned long long *counts) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;
  unsigned neg = 0, nan = 0, zero = 0;
  for (int jj = ty; jj < 32; jj += 8) {
    const int64_t i = i0 + tx, j = j0 + jj;
    float v = 0.f;
    if (i < n && j < m) {
      v = Xin[i + j * ldx];
      if (v < 0.f) neg++;
      if (v != v) nan++;
      if (v <= 0.f) {
        zero++;
        v = lambda;
      }
      Xc[i + j * n] = v;
    }
    tile[jj][tx] = v;
  }
  __syncthreads();
  for (int ii = ty; ii < 32; ii += 8) {
    const int64_t i = i0 + ii, j = j0 + tx;
    if (i < n && j < m) Xr[j + i * m] = tile[tx][ii];
  }
  if (neg) atomicAdd(&counts[0], (unsigned long long)neg);
  if (nan) atomicAdd(&counts[1], (unsigned long long)nan);
  if (zero) atomicAdd(&counts[2], (unsigned long long)zero);
}

__global__ __launch_bounds__(256) void fill_uniform_kernel(uint64_t seed, uint64_t offset, int64_t count, float *out) {
  const uint64_t key = nmfk_splitmix64(seed);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    out[i] = nmfk_uniform_keyed(key, offset + (uint64_t)i);
}

// ---------------------------------------------------------------------------------------------------
// clustersolutions: one workgroup; trials are sequential because every trial is compared against the
// running-sum centroids left by the previous ones (Clus:453-455, 484).
//   Hstack: nsol x (k x m), signal a of solution t = Hstack[t*k*m + a + j*k], j < m
//   work:   k*(m+1) floats (running-sum centroids, vector c at work[c*len + j])
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cluster_kernel(int k, int nsol, int m, const float *__restrict__ Hstack,
                                                      float *__restrict__ cent, int32_t *__restrict__ labels,
                                                      float *__restrict__ centroids, int32_t *needfix_out) {
  __shared__ float D[NMFK_MAX_K * NMFK_MAX_K];
  __shared__ int assign[NMFK_MAX_K];
  __shared__ float redv[256];
  __shared__ int redq[256];
  __shared__ int sh_needfix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t km = (int64_t)k * m;

  // zero-column guard (Clus:436-450): any all-zero signal => every vector gets a bias element of one
  if (tid == 0) sh_needfix = 0;
  __syncthreads();
  for (int v = wave; v < nsol * k; v += 4) {
    const int t = v / k, a = v - t * k;
    float s = 0.f;
    for (int j = lane; j < m; j += 64) s += Hstack[t * km + a + (int64_t)j * k];
    s = wave_sum_f(s);
    if (lane == 0 && s == 0.f) sh_needfix = 1;
  }
  __syncthreads();
  const int needfix = sh_needfix;
  const int len = needfix ? m + 1 : m;
  if (tid == 0) *needfix_out = needfix;

  for (int e = tid; e < k * len; e += 256) {
    const int c = e / len, j = e - c * len;
    cent[e] = (j < m) ? Hstack[c + (int64_t)j * k] : 1.0f;
  }
  for (int e = tid; e < k * nsol; e += 256) labels[e] = (e < k) ? e + 1 : 0;  // Clus:457-461
  __syncthreads();

  for (int t = 1; t < nsol; ++t) {
    const float *Ht = Hstack + t * km;
    // D[f + c*k] = cosine_dist(factor column f, centroid c); NaN -> 0 (Clus:467-473)
    for (int pq = wave; pq < k * k; pq += 4) {
      const int f = pq % k, c = pq / k;
      float ab = 0.f, a2 = 0.f, b2 = 0.f;
      for (int j = lane; j < m; j += 64) {
        const float x = Ht[f + (int64_t)j * k], y = cent[c * len + j];
        ab = fmaf(x, y, ab);
        a2 = fmaf(x, x, a2);
        b2 = fmaf(y, y, b2);
      }
      ab = wave_sum_f(ab);
      a2 = wave_sum_f(a2);
      b2 = wave_sum_f(b2);
      if (needfix) {  // bias element: x = 1, y = cent[c][m]
        const float y = cent[c * len + m];
        ab += y;
        a2 += 1.0f;
        b2 = fmaf(y, y, b2);
      }
      if (lane == 0) {
        float d = 1.0f - ab / (sqrtf(a2) * sqrtf(b2));
        d = (d != d) ? 0.0f : fmaxf(d, 0.0f);
        D[pq] = d;
      }
    }
    if (tid < NMFK_MAX_K) assign[tid] = -1;
    __syncthreads();
    // greedy one-to-one assignment, Julia's column-major first minimum (Clus:474-485)
    for (int round = 0; round < k; ++round) {
      float bv = __builtin_inff();
      int bq = 0x7fffffff;
      for (int q = tid; q < k * k; q += 256) {
        const float v = D[q];
        if (v < bv) {
          bv = v;
          bq = q;
        }
      }
      redv[tid] = bv;
      redq[tid] = bq;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
          const float v2 = redv[tid + o];
          const int q2 = redq[tid + o];
          if (v2 < redv[tid] || (v2 == redv[tid] && q2 < redq[tid])) {
            redv[tid] = v2;
            redq[tid] = q2;
          }
        }
        __syncthreads();
      }
      const int q = redq[0];
      const bool done = !(redv[0] < __builtin_inff());  // minimum(D) == Inf
      __syncthreads();
      if (done) break;
      const int f = q % k, c = q / k;
      if (tid == 0) {
        labels[f + t * k] = c + 1;
        assign[c] = f;
      }
      for (int e = tid; e < k; e += 256) {
        D[f + e * k] = __builtin_inff();
        D[e + c * k] = __builtin_inff();
      }
      __syncthreads();
    }
    // newClusterCenters[:, c] .+= W[:, f]  (Clus:484; aliased with the seeds the next trial compares to)
    for (int e = tid; e < k * len; e += 256) {
      const int c = e / len, j = e - c * len;
      const int f = assign[c];
      if (f >= 0) cent[e] += (j < m) ? Ht[f + (int64_t)j * k] : 1.0f;
    }
    __syncthreads();
  }
  // Clus:487-496 repairs, Clus:512-516 centroids ./= numTrials
  for (int e = tid; e < k * nsol; e += 256)
    if (labels[e] == 0) labels[e] = e % k + 1;
  for (int e = tid; e < k * m; e += 256) {
    const int c = e % k, j = e / k;
    centroids[e] = cent[c * len + j] / (float)nsol;
  }
}

// ---------------------------------------------------------------------------------------------------
// finalize: Z = zerostoepsilon(vcat(Ha...)) (Fin:52, Help:535-543), row p = a + t*k; norms; cosine distances
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zrows_kernel(int k, int nsol, int m, const float *__restrict__ Hstack,
                                                    float *__restrict__ Z, float *__restrict__ norms) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= k * nsol) return;
  const int t = p / k, a = p - t * k;
  const float e2 = 1.1920929e-07f * 1.1920929e-07f;
  float s = 0.f;
  for (int j = lane; j < m; j += 64) {
    float v = Hstack[(int64_t)t * k * m + a + (int64_t)j * k];
    v = (v < e2) ? e2 : v;
    Z[(int64_t)p * m + j] = v;
    s = fmaf(v, v, s);
  }
  s = wave_sum_f(s);
  if (lane == 0) norms[p] = sqrtf(s);
}

__global__ __launch_bounds__(256) void pairdist_kernel(int nT, int m, const float *__restrict__ Z,
                                                       const float *__restrict__ norms, float *__restrict__ D) {
  __shared__ float As[16][17], Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int p = blockIdx.y * 16 + ty, q = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int j0 = 0; j0 < m; j0 += 16) {
    const int pa = blockIdx.y * 16 + ty, qb = blockIdx.x * 16 + ty;
    As[ty][tx] = (pa < nT && j0 + tx < m) ? Z[(int64_t)pa * m + j0 + tx] : 0.f;
    Bs[ty][tx] = (qb < nT && j0 + tx < m) ? Z[(int64_t)qb * m + j0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) acc = fmaf(As[ty][jj], Bs[tx][jj], acc);
    __syncthreads();
  }
  if (p < nT && q < nT) {
    float d = 1.0f - acc / (norms[p] * norms[q]);
    d = (d != d) ? 0.0f : fmaxf(d, 0.0f);  // max(.,0) (Distances.cosine_dist); NaN -> 0 (Fin:53-54)
    if (p == q) d = 0.0f;                   // pairwise has a zero diagonal
    D[p + (int64_t)q * nT] = d;
  }
}

// Clustering.silhouettes(assignments, dists): one wavefront per point; for every cluster a masked
// wave-wide sum of the point's distance row (labels need not be permutations).
__global__ __launch_bounds__(256) void silhouette_kernel(int k, int nT, const float *__restrict__ D,
                                                         const int32_t *__restrict__ labels, float *__restrict__ psil) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= nT) return;
  const int own = labels[p] - 1;
  float mysum = 0.f;  // lane c keeps the sum / count of cluster c (k <= 64)
  int mycnt = 0;
  for (int c = 0; c < k; ++c) {
    float s = 0.f;
    int n = 0;
    for (int q = lane; q < nT; q += 64) {
      const bool in = (labels[q] - 1 == c);
      s += in ? D[p + (int64_t)q * nT] : 0.f;
      n += in ? 1 : 0;
    }
    s = wave_sum_f(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == c) {
      mysum = s;
      mycnt = n;
    }
  }
  const int cnt_own = __shfl(mycnt, own, 64);
  const float sum_own = __shfl(mysum, own, 64);
  float b = (lane < k && lane != own && mycnt > 0) ? mysum / (float)mycnt : __builtin_inff();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = fminf(b, __shfl_xor(b, o, 64));
  if (lane == 0) {
    float s = 0.f;
    if (cnt_own > 1) {
      const float a = sum_own / (float)(cnt_own - 1);
      s = (a < b) ? 1.0f - a / b : ((a > b) ? b / a - 1.0f : 0.0f);
    }
    psil[p] = (s != s) ? 0.0f : s;  // Fin:58
  }
}

__global__ __launch_bounds__(64) void cluster_mean_kernel(int nT, const int32_t *__restrict__ labels,
                                                          const float *__restrict__ psil, float *__restrict__ csil) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  int n = 0;
  for (int q = lane; q < nT; q += 64)
    if (labels[q] - 1 == c) {
      s += psil[q];
      n++;
    }
  s = wave_sum_f(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if (lane == 0) csil[c] = s / (float)n;  // Fin:66
}

// Fin:69-74: element-wise mean and corrected variance over the nsol members of each cluster
__global__ __launch_bounds__(256) void cluster_stats_kernel(int k, int nsol, int64_t len, int64_t sstride, int is_w,
                                                            const float *__restrict__ stack,
                                                            const int32_t *__restrict__ labels, float *__restrict__ mean,
                                                            float *__restrict__ var) {
  // is_w: stack element (x, a) of solution t at t*sstride + x + a*len (W: n x k); else at t*sstride + a + x*k (H)
  const int c = blockIdx.y;
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= len) return;
  float s = 0.f;
  int cnt = 0;
  for (int t = 0; t < nsol; ++t)
    for (int a = 0; a < k; ++a)
      if (labels[a + t * k] - 1 == c) {
        s += is_w ? stack[t * sstride + x + a * len] : stack[t * sstride + a + x * k];
        cnt++;
      }
  const float mu = s / (float)cnt;
  float s2 = 0.f;
  for (int t = 0; t < nsol; ++t)
    for (int a = 0; a < k; ++a)
      if (labels[a + t * k] - 1 == c) {
        const float d = (is_w ? stack[t * sstride + x + a * len] : stack[t * sstride + a + x * k]) - mu;
        s2 = fmaf(d, d, s2);
      }
  const int64_t o = is_w ? x + c * len : c + x * k;
  mean[o] = mu;
  var[o] = s2 / (float)(cnt - 1);
}

// normnan(X - W*H) (Help:226-228) for caller-supplied fp32 factors (re-checks at Exec:603, 664-667, 212-222).
// W: n x k column-major, H: k x m column-major.  One partial per workgroup, summed by the host in order.
__global__ __launch_bounds__(256) void frob_kernel(const float *__restrict__ Xc, int n, int m, int k,
                                                   const float *__restrict__ W, const float *__restrict__ H,
                                                   double *__restrict__ partial) {
  __shared__ double sh[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n;
  const int ic = valid ? i : 0;
  double s = 0.0;
  for (int j = 0; j < m; ++j) {
    float p = 0.f;
    for (int c = 0; c < k; ++c) p = fmaf(W[ic + (int64_t)c * n], H[c + (int64_t)j * k], p);
    const float x = Xc[ic + (int64_t)j * n];
    const float e = x - p;
    if (valid && e == e) s += (double)e * (double)e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

}  // namespace

void nmfk_launch_frob(const float *Xc, int n, int m, int k, const float *W, const float *H, double *partial,
                      hipStream_t s) {
  hipLaunchKernelGGL(frob_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Xc, n, m, k, W, H, partial);
}

void nmfk_launch_preprocess(const float *Xin, int64_t ldx, int64_t n, int64_t m, float lambda, float *Xc, float *Xr,
                            unsigned long long *counts, hipStream_t s) {
  dim3 grid((unsigned)((n + 31) / 32), (unsigned)((m + 31) / 32));
  hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, s, Xin, ldx, n, m, lambda, Xc, Xr, counts);
}

void nmfk_launch_fill_uniform(uint64_t seed, uint64_t offset, int64_t count, float *out, hipStream_t s) {
  int64_t blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, s, seed, offset, count, out);
}

void nmfk_launch_cluster(int k, int nsol, int m, const float *Hstack, float *work, int32_t *labels, float *centroids,
                         int32_t *needfix, hipStream_t s) {
  hipLaunchKernelGGL(cluster_kernel, dim3(1), dim3(256), 0, s, k, nsol, m, Hstack, work, labels, centroids, needfix);
}

void nmfk_launch_silhouette(int k, int nsol, int m, const float *Hstack, const int32_t *labels, float *Z, float *norms,
                            float *D, float *psil, float *csil, hipStream_t s) {
  const int nT = k * nsol;
  hipLaunchKernelGGL(zrows_kernel, dim3((nT + 3) / 4), dim3(256), 0, s, k, nsol, m, Hstack, Z, norms);
  hipLaunchKernelGGL(pairdist_kernel, dim3((nT + 15) / 16, (nT + 15) / 16), dim3(256), 0, s, nT, m, Z, norms, D);
  hipLaunchKernelGGL(silhouette_kernel, dim3((nT + 3) / 4), dim3(256), 0, s, k, nT, D, labels, psil);
  hipLaunchKernelGGL(cluster_mean_kernel, dim3(k), dim3(64), 0, s, nT, labels, psil, csil);
}

void nmfk_launch_cluster_stats(int k, int nsol, int n, int m, const float *Wstack, const float *Hstack,
                               const int32_t *labels, float *Wmean, float *Hmean, float *Wvar, float *Hvar,
                               hipStream_t s) {
  hipLaunchKernelGGL(cluster_stats_kernel, dim3((n + 255) / 256, k), dim3(256), 0, s, k, nsol, (int64_t)n,
                     (int64_t)n * k, 1, Wstack, labels, Wmean, Wvar);
  hipLaunchKernelGGL(cluster_stats_kernel, dim3((m + 255) / 256, k), dim3(256), 0, s, k, nsol, (int64_t)m,
                     (int64_t)m * k, 0, Hstack, labels, Hmean, Hvar);
}
