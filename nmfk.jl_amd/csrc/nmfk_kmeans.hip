// robustkmeans(X, k, repeats)  (src/NMFkCluster.jl:172-246) for gfx950.
//
// The reference loops `repeats` (default 1000) independent Clustering.kmeans runs on the CPU and keeps the cheapest.
// Here every repeat is one workgroup and all repeats run concurrently.  The arithmetic of one run is the one restated
// in oracle/nmfk_oracle.c (k-means++ seeding with squared Euclidean distances, Lloyd iterations with cosine
// distances, running-sum centre update in column order, empty-cluster re-pick, |objective change| < tol) and is
// kept in the SAME operation order (sequential dot products and sums, fp contraction off for this file) so that a
// repeat reproduces the oracle's assignments bit for bit; the random draws come from the shared counter-based
// generator.  Work per repeat is tiny (n*k*d flops per iteration); the sequential parts (cumulative sums of the
// weighted sampling, the objective) run on one thread of the workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmfk_hip.h"
#include "nmfk_common.h"
#include "nmfk_rng.h"

#pragma clang fp contract(off)

namespace {

constexpr int KM_THREADS = 256;

__device__ __forceinline__ double km_uniform(uint64_t key, uint64_t idx) {
  return (double)nmfk_uniform_keyed(key, idx);  // odd 24-bit integer * 2^-24: the same value in fp32 and fp64
}

__device__ __forceinline__ float km_sqeuclid(const float *a, const float *b, int d) {
  float s = 0.f;
  for (int i = 0; i < d; ++i) {
    const float v = a[i] - b[i];
    s += v * v;
  }
  return s;
}

// Distances.cosine_dist: max(1 - <a,b>/(|a||b|), 0), NaN propagates (same expression order as the oracle's)
__device__ __forceinline__ float km_cosine(const float *a, const float *b, int d) {
  float ab = 0.f, a2 = 0.f, b2 = 0.f;
  for (int i = 0; i < d; ++i) {
    const float x = a[i], y = b[i];
    ab += x * y;
    a2 += x * x;
    b2 += y * y;
  }
  const float dd = 1.0f - ab / (sqrtf(a2) * sqrtf(b2));
  return dd > 0.f ? dd : (dd == dd ? 0.f : dd);
}

// StatsBase.sample(Weights(w)) with u in (0,1): one thread, sequential double sums (order = the oracle's)
__device__ int km_wsample(const float *w, int n, double u) {
  double tot = 0;
  for (int i = 0; i < n; ++i) tot += (double)w[i];
  const double t = u * tot;
  int i = 0;
  double cw = (double)w[0];
  while (cw < t && i < n - 1) {
    ++i;
    cw += (double)w[i];
  }
  return i;
}

struct KmArgs {
  const float *X;  // d x n, columns = samples
  int d, n, k, maxiter;
  double tol;
  uint64_t seed;
  // per repeat (stride = index of the repeat)
  int32_t *assign;  // [repeats][n] 0-based
  float *costs;     // [repeats][n]
  float *work;      // [repeats][n] k-means++ / re-pick costs
  float *centers;   // [repeats][d*k]
  int32_t *counts;  // [repeats][k]
  double *total;    // [repeats]
  int32_t *iters;   // [repeats]
  int32_t *conv;    // [repeats]
};

__global__ __launch_bounds__(KM_THREADS) void kmeans_kernel(KmArgs g) {
  extern __shared__ float cen[];  // d*k centres, then k counts / flags
  const int r = blockIdx.x, tid = threadIdx.x;
  const int d = g.d, n = g.n, k = g.k;
  int32_t *cnt = (int32_t *)(cen + d * k);
  int32_t *upd = cnt + k;
  int32_t *unused = upd + k;
  __shared__ int sh_p, sh_nun, sh_stop;
  __shared__ double sh_objv;
  const float *X = g.X;
  int32_t *assign = g.assign + (size_t)r * n;
  float *costs = g.costs + (size_t)r * n;
  float *mc = g.work + (size_t)r * n;
  const uint64_t key = nmfk_splitmix64(g.seed + (uint64_t)r);
  uint64_t draw = 0;  // only thread 0's copy is used

  // ---- k-means++ seeding (squared Euclidean) ----
  if (tid == 0) {
    int p = (int)(km_uniform(key, draw++) * (double)n);
    sh_p = p > n - 1 ? n - 1 : p;
  }
  __syncthreads();
  for (int i = tid; i < d; i += KM_THREADS) cen[i] = X[i + (size_t)sh_p * d];
  if (k > 1) {
    for (int j = tid; j < n; j += KM_THREADS) mc[j] = j == sh_p ? 0.f : km_sqeuclid(X + (size_t)j * d, X + (size_t)sh_p * d, d);
    for (int c = 1; c < k; ++c) {
      __syncthreads();
      if (tid == 0) sh_p = km_wsample(mc, n, km_uniform(key, draw++));
      __syncthreads();
      const int p = sh_p;
      for (int i = tid; i < d; i += KM_THREADS) cen[i + c * d] = X[i + (size_t)p * d];
      for (int j = tid; j < n; j += KM_THREADS) {
        const float v = km_sqeuclid(X + (size_t)j * d, X + (size_t)p * d, d);
        float m = mc[j];
        if (v < m) m = v;
        mc[j] = j == p ? 0.f : m;
      }
    }
  }
  __syncthreads();

  int it = 0;
  double prev = 0;
  for (int pass = 0;; ++pass) {  // pass 0 = initial assignment
    if (pass > 0) {
      ++it;
      // update_centers!: running sum of the member columns in column order, then / count (affected clusters only)
      for (int e = tid; e < d * k; e += KM_THREADS) {
        const int c = e / d, i = e - c * d;
        if (!upd[c]) continue;
        float s = 0.f;
        int w = 0;
        for (int j = 0; j < n; ++j)
          if (assign[j] == c) {
            s = w > 0 ? s + X[i + (size_t)j * d] : X[i + (size_t)j * d];
            ++w;
          }
        cen[e] = s / (float)w;
      }
      __syncthreads();
      if (sh_nun > 0) {  // repick_unused_centers
        for (int j = tid; j < n; j += KM_THREADS) mc[j] = costs[j];
        const int nun = sh_nun;
        for (int q = 0; q < nun; ++q) {
          __syncthreads();
          if (tid == 0) sh_p = km_wsample(mc, n, km_uniform(key, draw++));
          __syncthreads();
          const int p = sh_p, c = unused[q];
          for (int i = tid; i < d; i += KM_THREADS) cen[i + c * d] = X[i + (size_t)p * d];
          for (int j = tid; j < n; j += KM_THREADS) {
            const float v = km_cosine(X + (size_t)p * d, X + (size_t)j * d, d);
            float m = j == p ? 0.f : mc[j];
            if (v < m) m = v;
            mc[j] = m;
          }
        }
        __syncthreads();
      }
    }
    // update_assignments!
    for (int c = tid; c < k; c += KM_THREADS) {
      cnt[c] = 0;
      upd[c] = pass == 0;
    }
    __syncthreads();
    for (int j = tid; j < n; j += KM_THREADS) {
      const float *x = X + (size_t)j * d;
      int a = 0;
      float cm = km_cosine(cen, x, d);
      for (int c = 1; c < k; ++c) {
        const float ci = km_cosine(cen + c * d, x, d);
        if (ci < cm) {
          a = c;
          cm = ci;
        }
      }
      if (pass == 0) {
        assign[j] = a;
      } else if (assign[j] != a) {
        atomicOr(&upd[a], 1);
        atomicOr(&upd[assign[j]], 1);
        assign[j] = a;
      }
      costs[j] = cm;
      atomicAdd(&cnt[a], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int nun = 0;
      for (int c = 0; c < k; ++c)
        if (cnt[c] == 0) {
          unused[nun++] = c;
          upd[c] = 0;
        }
      sh_nun = nun;
      double objv = 0;
      for (int j = 0; j < n; ++j) objv += (double)costs[j];
      int converged = 0;
      if (pass > 0) {
        const double ch = objv - prev;
        if (!(ch > g.tol) && (k == 1 || fabs(ch) < g.tol)) converged = 1;
      }
      prev = objv;
      sh_objv = objv;
      sh_stop = converged ? 2 : (it >= g.maxiter ? 1 : 0);
    }
    __syncthreads();
    if (sh_stop) break;
  }
  for (int e = tid; e < d * k; e += KM_THREADS) g.centers[(size_t)r * d * k + e] = cen[e];
  for (int c = tid; c < k; c += KM_THREADS) g.counts[(size_t)r * k + c] = cnt[c];
  if (tid == 0) {
    g.total[r] = sh_objv;
    g.iters[r] = it;
    g.conv[r] = sh_stop == 2;
  }
}

// Clustering.silhouettes on dists = pairwise(CosineDist(), zerostoepsilon(X); dims=2) (Clus:204-213): one thread per
// point, sequential over the other points (the oracle's order).  assign 1-based, k <= NMFK_MAX_K.
__global__ __launch_bounds__(KM_THREADS) void point_silhouette_kernel(const float *X, int d, int n, const int32_t *assign,
                                                                      const int32_t *cnt, int k, float *Z, float *sil) {
  const int i = blockIdx.x * KM_THREADS + threadIdx.x;
  if (i >= n) return;
  float s[NMFK_MAX_K];
  for (int c = 0; c < k; ++c) s[c] = 0.f;
  const float *zi = Z + (size_t)i * d;
  for (int j = 0; j < n; ++j)
    if (j != i) s[assign[j] - 1] += km_cosine(zi, Z + (size_t)j * d, d);
  const int ci = assign[i] - 1;
  if (cnt[ci] <= 1) {
    sil[i] = 0.f;
    return;
  }
  const float a = s[ci] / (float)(cnt[ci] - 1);
  float b = 0.f;
  int have = 0;
  for (int c = 0; c < k; ++c) {
    if (c == ci || cnt[c] == 0) continue;
    const float v = s[c] / (float)cnt[c];
    if (!have || v < b) b = v;
    have = 1;
  }
  const float mx = a > b ? a : b;
  sil[i] = have ? (b - a) / mx : 0.f;
}

__global__ void zerostoeps_kernel(const float *X, size_t cnt, float *Z) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const float e2 = 1.4210854715202004e-14f;  // eps(Float32)^2 (Help:535-543)
  const float v = X[i];
  Z[i] = v < e2 ? e2 : v;
}

}  // namespace

void nmfk_launch_kmeans(const float *X, int d, int n, int k, int repeats, int maxiter, double tol, uint64_t seed,
                        int32_t *assign, float *costs, float *work, float *centers, int32_t *counts, double *total,
                        int32_t *iters, int32_t *conv, hipStream_t s) {
  KmArgs g{X, d, n, k, maxiter, tol, seed, assign, costs, work, centers, counts, total, iters, conv};
  const size_t lds = sizeof(float) * (size_t)d * k + sizeof(int32_t) * 3 * (size_t)k;
  hipLaunchKernelGGL(kmeans_kernel, dim3(repeats), dim3(KM_THREADS), lds, s, g);
}

void nmfk_launch_point_silhouettes(const float *X, int d, int n, const int32_t *assign, const int32_t *cnt, int k, float *Z,
                                   float *sil, hipStream_t s) {
  const size_t tot = (size_t)d * n;
  hipLaunchKernelGGL(zerostoeps_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, X, tot, Z);
  hipLaunchKernelGGL(point_silhouette_kernel, dim3((n + KM_THREADS - 1) / KM_THREADS), dim3(KM_THREADS), 0, s, X, d, n,
                     assign, cnt, k, Z, sil);
}
