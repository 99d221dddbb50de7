/*
 * nmfk_oracle.c -- CPU restatement of the NMFk.jl `execute(...; method=:simple)` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product path (nmfk.jl_amd/) never links or calls it.
 *
 * Parity status: the reference is pure Julia and cannot run in the build container (no julia).
 * This restatement is pinned against every known-answer the reference holds for the path
 * (tests/test_oracle_golden.py): test/test_cluster_unit.jl:36-54, test/test_execute_smoke.jl:6-32,
 * test/test_normalize.jl:44-55, the blind-source-separation notebook's printed X and results
 * (notebooks/blind_source_separation/blind_source_separation.md:161-181, 219-264) and Readme.md:120-134.
 * The arithmetic that lives in un-vendored third-party Julia packages is restated from the packages'
 * published definitions and cross-checked against scipy / scikit-learn in tests/test_oracle_units.py:
 *   Distances.jl (compat 0.8-0.11, Project.toml:63): cosine_dist(a,b) = max(1 - <a,b>/(|a||b|), 0)
 *   Clustering.jl (compat 0.14-0.15, Project.toml:55): silhouettes(assignments, dists)
 * No reference test asserts a silhouette value => that part is "parity unpinned" (see DESIGN.md).
 *
 * All matrices are column-major (Julia layout).  The inner MU loop runs in Float64 exactly as the
 * reference does on its default path (W = rand(n,k), H = rand(k,m) are Float64 even for Float32 X,
 * src/NMFkMultiplicative.jl:38,48); `tbits` = 32/64 states the element type T of X, which only
 * matters where the reference stores into T-typed containers (imputed entries Mult:72, WBig/HBig/objvalue
 * Exec:529-531, and everything downstream: clustering, silhouettes run in T).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

typedef struct {
  double tol;          /* Mult:24 tol=1e-19 (Exec:729 passes 1e-19)            */
  double tolOF;        /* Mult:24 tolOF=1e-3                                   */
  double lambda;       /* Mult:24 lambda=1e-32                                 */
  double weight;       /* scalar weight (Mult:74); array weights: next rows    */
  int64_t maxiter;     /* Exec:729 maxiter=10000                               */
  int32_t maxreattempts; /* 2  */
  int32_t maxbaditers;   /* 10 */
  int32_t stopconv;      /* 1000 */
  int32_t Wfixed;
  int32_t Hfixed;
  int32_t tbits;       /* 32 or 64: element type T of X                        */
  int32_t nthreads;    /* OpenMP threads (results do not depend on it)         */
} nmfk_or_params;

enum { STOP_MAXITER = 1, STOP_STAGNATION = 2, STOP_TOL = 3, STOP_CONSISTENCY = 4 };

static inline double round_T(double v, int tbits) { return tbits == 32 ? (double)(float)v : v; }

EXPORT int nmfk_or_version(void) { return 1; }

EXPORT int nmfk_or_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Portable counter-based uniform generator shared bit-for-bit with the HIP library
 * (nmfk.jl_amd/csrc/nmfk_rng.h).  Stands in for Julia's rand(n,k) (Mult:38,48), whose stream cannot be
 * reproduced without Julia.  Value = odd 24-bit integer * 2^-24 in (0,1): exact in fp32 and fp64. */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double nmfk_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = splitmix64(splitmix64(seed) ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
  uint32_t b = (uint32_t)(z >> 41); /* 23 bits */
  return (double)(2u * b + 1u) * (1.0 / 16777216.0);
}
/* W = rand(n,k) first, then H = rand(k,m)  (Mult:38,48 order) */
EXPORT void nmfk_or_init(uint64_t seed, int64_t n, int64_t m, int64_t k, double *W, double *H) {
  for (int64_t i = 0; i < n * k; i++) W[i] = nmfk_uniform(seed, (uint64_t)i);
  for (int64_t i = 0; i < k * m; i++) H[i] = nmfk_uniform(seed, (uint64_t)(n * k + i));
}
EXPORT void nmfk_or_uniform_fill(uint64_t seed, uint64_t offset, int64_t count, double *out) {
  for (int64_t i = 0; i < count; i++) out[i] = nmfk_uniform(seed, offset + (uint64_t)i);
}

/* NMFpreprocessing!  src/NMFkMultiplicative.jl:3-22.
 * returns -1 when minimum(X) < 0 (ErrorException "All matrix entries must be nonnegative!").
 * izero = X .<= 0 is taken BEFORE the NaN replacement (NaN compares false). */
EXPORT int nmfk_or_preprocess(double *X, int64_t n, int64_t m, double lambda, uint8_t *inan, uint8_t *izero) {
  int64_t N = n * m;
  for (int64_t i = 0; i < N; i++)
    if (X[i] < 0) return -1;
  for (int64_t i = 0; i < N; i++) {
    izero[i] = (X[i] <= 0) ? 1 : 0;
    if (izero[i]) X[i] = lambda;
  }
  for (int64_t i = 0; i < N; i++) {
    inan[i] = isnan(X[i]) ? 1 : 0;
    if (inan[i]) X[i] = lambda;
  }
  return 0;
}

/* sum((((X - W*H) .* weight)[.!inan]).^2)   Mult:74,125.  Per-column partials summed in column order. */
static double sse_masked(const double *X, const uint8_t *inan, const double *Wt /*k x n*/, const double *H,
                         int64_t n, int64_t m, int64_t k, double weight, const double *warr /* n x m or NULL */,
                         double *colpart) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < m; j++) {
    const double *h = H + j * k;
    double s = 0;
    for (int64_t i = 0; i < n; i++) {
      if (inan[i + j * n]) continue;
      const double *w = Wt + i * k;
      double p = 0;
      for (int64_t a = 0; a < k; a++) p += w[a] * h[a];
      double e = (X[i + j * n] - p) * (warr ? weight * warr[i + j * n] : weight);
      s += e * e;
    }
    colpart[j] = s;
  }
  double t = 0;
  for (int64_t j = 0; j < m; j++) t += colpart[j];
  return t;
}

/* NMFmultiplicative  src/NMFkMultiplicative.jl:24-127  (dense method; scalar weight).
 * X: n x m col-major, values of type T held in doubles; mutated during the loop and restored on exit
 * (Mult:123-124).  W (n x k), H (k x m): initial values in, final values out.  Returns 0, or -1 for a
 * negative entry.  The state machine is §3.2 of SURVEY.md, line by line. */
EXPORT int nmfk_or_multiplicative_ex(double *X, int64_t n, int64_t m, int64_t k, const nmfk_or_params *P, double *W,
                                     double *H, double *sse_out, int64_t *iters_out, int32_t *reason_out,
                                     int32_t *nchecks_out, double *objtrace /* may be NULL; len maxiter/10 */,
                                     const double *warr /* weight array n x m (Mult:74) or NULL */,
                                     const double *normvec /* normalizevector, length n (Mult:27-31) or NULL */) {
#ifdef _OPENMP
  omp_set_num_threads(P->nthreads > 0 ? P->nthreads : 1);
#endif
  int64_t N = n * m;
  uint8_t *inan = (uint8_t *)malloc(N), *izero = (uint8_t *)malloc(N);
  if (nmfk_or_preprocess(X, n, m, P->lambda, inan, izero) != 0) {
    free(inan);
    free(izero);
    return -1;
  }
  int64_t nnan = 0;
  for (int64_t i = 0; i < N; i++) nnan += inan[i];
  if (normvec) /* Mult:27-28  X ./= normalizevector (rows) */
    for (int64_t j = 0; j < m; j++)
      for (int64_t i = 0; i < n; i++) X[i + j * n] = round_T(X[i + j * n] / normvec[i], P->tbits);

  double *Wt = (double *)malloc(sizeof(double) * n * k);   /* k x n: row i of W contiguous */
  double *Xt = (double *)malloc(sizeof(double) * N);        /* m x n: row i of X contiguous */
  double *Hn = (double *)malloc(sizeof(double) * k * m);
  double *cs = (double *)malloc(sizeof(double) * k), *rs = (double *)malloc(sizeof(double) * k);
  double *colpart = (double *)malloc(sizeof(double) * m);
  int64_t *index = (int64_t *)malloc(sizeof(int64_t) * m), *canon = (int64_t *)malloc(sizeof(int64_t) * m);
  int64_t *canon_old = (int64_t *)malloc(sizeof(int64_t) * m), *first = (int64_t *)malloc(sizeof(int64_t) * k);
  for (int64_t i = 0; i < n; i++)
    for (int64_t a = 0; a < k; a++) Wt[a + i * k] = W[i + a * n];
  for (int64_t j = 0; j < m; j++)
    for (int64_t i = 0; i < n; i++) Xt[j + i * m] = X[i + j * n];

  /* consold = falses(m,m) (Mult:57): represented as "no partition yet"; the first comparison always
   * differs because cons has a true diagonal. */
  int have_old = 0;
  int64_t inc = 0, iters = 0;
  int32_t baditers = 0, reattempts = 0, reason = 0, nchecks = 0;
  double best = INFINITY;

  while (iters < P->maxiter && baditers < P->maxbaditers && reattempts < P->maxreattempts) { /* Mult:64 */
    iters += 1;
    if (!P->Hfixed) { /* Mult:67  H = H .* (W' * (X ./ (W*H))) ./ sum(W;dims=1)' */
      for (int64_t a = 0; a < k; a++) {
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += W[i + a * n];
        cs[a] = s;
      }
#pragma omp parallel for schedule(static)
      for (int64_t j = 0; j < m; j++) {
        const double *h = H + j * k;
        double acc[64];
        for (int64_t a = 0; a < k; a++) acc[a] = 0;
        for (int64_t i = 0; i < n; i++) {
          const double *w = Wt + i * k;
          double p = 0;
          for (int64_t a = 0; a < k; a++) p += w[a] * h[a];
          double q = X[i + j * n] / p;
          for (int64_t a = 0; a < k; a++) acc[a] += w[a] * q;
        }
        for (int64_t a = 0; a < k; a++) Hn[a + j * k] = h[a] * acc[a] / cs[a];
      }
      memcpy(H, Hn, sizeof(double) * k * m);
    }
    if (!P->Wfixed) { /* Mult:70  W = W .* ((X ./ (W*H)) * H') ./ sum(H;dims=2)'   (new H) */
      for (int64_t a = 0; a < k; a++) rs[a] = 0;
      for (int64_t j = 0; j < m; j++)
        for (int64_t a = 0; a < k; a++) rs[a] += H[a + j * k];
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) {
        double *w = Wt + i * k;
        double acc[64];
        for (int64_t a = 0; a < k; a++) acc[a] = 0;
        for (int64_t j = 0; j < m; j++) {
          const double *h = H + j * k;
          double p = 0;
          for (int64_t a = 0; a < k; a++) p += w[a] * h[a];
          double q = Xt[j + i * m] / p;
          for (int64_t a = 0; a < k; a++) acc[a] += q * h[a];
        }
        for (int64_t a = 0; a < k; a++) {
          w[a] = w[a] * acc[a] / rs[a];
          W[i + a * n] = w[a];
        }
      }
    }
    if (nnan > 0) { /* Mult:72  X[inan] = (W*H)[inan], stored into X::Matrix{T} */
#pragma omp parallel for schedule(static)
      for (int64_t j = 0; j < m; j++)
        for (int64_t i = 0; i < n; i++)
          if (inan[i + j * n]) {
            double p = 0;
            for (int64_t a = 0; a < k; a++) p += Wt[a + i * k] * H[a + j * k];
            p = round_T(p, P->tbits);
            X[i + j * n] = p;
            Xt[j + i * m] = p;
          }
    }
    if (iters % 10 == 0) { /* Mult:73-117 */
      double obj = sse_masked(X, inan, Wt, H, n, m, k, P->weight, warr, colpart);
      if (objtrace) objtrace[nchecks] = obj;
      nchecks++;
      if (obj < P->tol) { /* Mult:75-78 */
        reason = STOP_TOL;
        break;
      }
      if (obj < best) { /* Mult:79-89 */
        if ((best - obj) < P->tolOF)
          baditers += 1;
        else
          baditers = 0;
        best = obj;
      } else {
        baditers += 1;
      }
      if (baditers >= P->maxbaditers) { /* Mult:90-98 */
        reattempts += 1;
        baditers = 0;
      }
      const double eps = 2.220446049250313e-16; /* eps() is Float64 eps regardless of T, Mult:99-100 */
      /* Julia's max propagates NaN: max(NaN, eps()) = NaN, so a NaN factor entry stays NaN */
      for (int64_t i = 0; i < k * m; i++) H[i] = H[i] < eps ? eps : H[i];
      for (int64_t i = 0; i < n * k; i++) {
        W[i] = W[i] < eps ? eps : W[i];
      }
      for (int64_t i = 0; i < n; i++)
        for (int64_t a = 0; a < k; a++) Wt[a + i * k] = W[i + a * n];
      /* Mult:101-116: cons[i,j] = (index[i]==index[j]); consdiff==0  <=>  same partition of the columns.
       * Canonical form: label every column by the first column of its class. */
      for (int64_t a = 0; a < k; a++) first[a] = -1;
      for (int64_t q = 0; q < m; q++) {
        int64_t am = 0; /* argmin: first minimum; the first NaN wins (Julia's argmin) */
        for (int64_t a = 1; a < k; a++) {
          if (isnan(H[am + q * k])) break;
          if (isnan(H[a + q * k]) || H[a + q * k] < H[am + q * k]) am = a;
        }
        index[q] = am;
        if (first[am] < 0) first[am] = q;
        canon[q] = first[am];
      }
      int same = have_old;
      if (have_old)
        for (int64_t q = 0; q < m; q++)
          if (canon[q] != canon_old[q]) {
            same = 0;
            break;
          }
      if (same)
        inc += 1;
      else
        inc = 0;
      if (inc > P->stopconv) {
        reason = STOP_CONSISTENCY;
        break;
      }
      memcpy(canon_old, canon, sizeof(int64_t) * m);
      have_old = 1;
    }
  }
  if (reason == 0) /* the loop guard failed (Mult:64) */
    reason = (reattempts >= P->maxreattempts || baditers >= P->maxbaditers) ? STOP_STAGNATION : STOP_MAXITER;

  if (normvec) { /* Mult:119-122  X .*= normalizevector; W .*= normalizevector */
    for (int64_t j = 0; j < m; j++)
      for (int64_t i = 0; i < n; i++) X[i + j * n] = round_T(X[i + j * n] * normvec[i], P->tbits);
    for (int64_t a = 0; a < k; a++)
      for (int64_t i = 0; i < n; i++) {
        W[i + a * n] *= normvec[i];
        Wt[a + i * k] = W[i + a * n];
      }
  }
  /* Mult:123-126: restore X, final SSE on the restored X */
  for (int64_t i = 0; i < N; i++) {
    if (izero[i]) X[i] = 0;
    if (inan[i]) X[i] = NAN;
  }
  *sse_out = sse_masked(X, inan, Wt, H, n, m, k, P->weight, warr, colpart);
  *iters_out = iters;
  *reason_out = reason;
  if (nchecks_out) *nchecks_out = nchecks;
  free(inan); free(izero); free(Wt); free(Xt); free(Hn); free(cs); free(rs); free(colpart);
  free(index); free(canon); free(canon_old); free(first);
  return 0;
}

EXPORT int nmfk_or_multiplicative(double *X, int64_t n, int64_t m, int64_t k, const nmfk_or_params *P, double *W,
                                  double *H, double *sse_out, int64_t *iters_out, int32_t *reason_out,
                                  int32_t *nchecks_out, double *objtrace) {
  return nmfk_or_multiplicative_ex(X, n, m, k, P, W, H, sse_out, iters_out, reason_out, nchecks_out, objtrace, NULL, NULL);
}

/* normnan(X - W*H)  (src/NMFkHelpers.jl:226-228 via Exec:791-792): Frobenius norm over non-NaN entries. */
EXPORT double nmfk_or_frobenius(const double *X, int64_t n, int64_t m, int64_t k, const double *W, const double *H) {
  double t = 0;
  for (int64_t j = 0; j < m; j++) {
    double s = 0;
    for (int64_t i = 0; i < n; i++) {
      double x = X[i + j * n];
      if (isnan(x)) continue;
      double p = 0;
      for (int64_t a = 0; a < k; a++) p += W[i + a * n] * H[a + j * k];
      if (isnan(p)) continue;
      s += (x - p) * (x - p);
    }
    t += s;
  }
  return sqrt(t);
}

/* execute_singlerun_compute, :simple branch  src/NMFkExecute.jl:729-807:
 * NMFmultiplicative -> objvalue = normnan(X - W*H) (Exec:791-792) -> rows of H sum to 1 (Exec:801-803,
 * skipped when modifymatrices=false i.e. Wfixed/Hfixed given, Exec:486-489). */
EXPORT int nmfk_or_singlerun(double *X, int64_t n, int64_t m, int64_t k, const nmfk_or_params *P, int32_t modifymatrices,
                             double *W, double *H, double *objvalue, double *sse_out, int64_t *iters_out,
                             int32_t *reason_out, const double *warr, const double *normvec) {
  int rc = nmfk_or_multiplicative_ex(X, n, m, k, P, W, H, sse_out, iters_out, reason_out, NULL, NULL, warr, normvec);
  if (rc) return rc;
  *objvalue = nmfk_or_frobenius(X, n, m, k, W, H);
  if (modifymatrices) {
    for (int64_t a = 0; a < k; a++) {
      double total = 0;
      for (int64_t j = 0; j < m; j++) total += H[a + j * k];
      for (int64_t i = 0; i < n; i++) W[i + a * n] *= total;
      for (int64_t j = 0; j < m; j++) H[a + j * k] /= total;
    }
  }
  return 0;
}

/* ---- clustering / silhouettes, computed in T (Exec:529-531 store Matrix{T}; Clus:463) ---- */
#define DEF_COSINE(NAME, T, SQRT)                                                      \
  static inline T NAME(const T *a, int64_t sa, const T *b, int64_t sb, int64_t len) { \
    T ab = 0, a2 = 0, b2 = 0;                                                          \
    for (int64_t i = 0; i < len; i++) {                                                \
      T x = a[i * sa], y = b[i * sb];                                                  \
      ab += x * y;                                                                     \
      a2 += x * x;                                                                     \
      b2 += y * y;                                                                     \
    }                                                                                  \
    T d = (T)1 - ab / (SQRT(a2) * SQRT(b2));                                           \
    return d > 0 ? d : (d == d ? (T)0 : d); /* max(.,0); NaN propagates */             \
  }
DEF_COSINE(cosine_f32, float, sqrtf)
DEF_COSINE(cosine_f64, double, sqrt)

/* clustersolutions(factors, clusterWmatrix=false)  src/NMFkCluster.jl:425-517.
 * F: R matrices k x m (each H of one solution, col-major), stacked [r][a + j*k]; solutions already sorted
 * by objective by the caller (Exec:623).  labels: k x R col-major, 1-based.  centroids: k x m.
 * The bias-row zero fix (Clus:436-450), the aliasing of centSeeds/newClusterCenters (Clus:453-455), the
 * NaN->0 rule (Clus:473) and Julia's column-major first-minimum argmin (Clus:476) are reproduced. */
#define DEF_CLUSTER(NAME, T, COS)                                                                           \
  EXPORT int NAME(const T *F, int64_t R, int64_t k, int64_t m, int32_t *labels, T *centroids) {             \
    int64_t len = m;                                                                                        \
    int needfix = 0;                                                                                        \
    for (int64_t r = 0; r < R && !needfix; r++)                                                             \
      for (int64_t a = 0; a < k; a++) {                                                                     \
        T s = 0;                                                                                            \
        for (int64_t j = 0; j < m; j++) s += F[r * k * m + a + j * k];                                      \
        if (s == 0) { needfix = 1; break; }                                                                 \
      }                                                                                                     \
    if (needfix) len = m + 1;                                                                               \
    /* working copies: signal a of solution r as a contiguous vector of length len */                      \
    T *V = (T *)malloc(sizeof(T) * R * k * len);                                                            \
    for (int64_t r = 0; r < R; r++)                                                                         \
      for (int64_t a = 0; a < k; a++) {                                                                     \
        for (int64_t j = 0; j < m; j++) V[(r * k + a) * len + j] = F[r * k * m + a + j * k];                \
        if (needfix) V[(r * k + a) * len + m] = (T)1;                                                       \
      }                                                                                                     \
    T *cent = V; /* factors[1] aliased: running sums live in solution 1's own storage */                   \
    T *D = (T *)malloc(sizeof(T) * k * k);                                                                  \
    for (int64_t i = 0; i < k * R; i++) labels[i] = 0;                                                      \
    for (int64_t a = 0; a < k; a++) labels[a] = (int32_t)(a + 1);                                           \
    for (int64_t t = 1; t < R; t++) {                                                                       \
      const T *Wt = V + t * k * len;                                                                        \
      for (int64_t c = 0; c < k; c++)                                                                       \
        for (int64_t f = 0; f < k; f++) {                                                                   \
          T d = COS(Wt + f * len, 1, cent + c * len, 1, len);                                               \
          D[f + c * k] = (d != d) ? (T)0 : d;                                                               \
        }                                                                                                   \
      for (;;) {                                                                                            \
        int64_t best = -1;                                                                                  \
        T bv = (T)INFINITY;                                                                                 \
        for (int64_t q = 0; q < k * k; q++)                                                                 \
          if (D[q] < bv) { bv = D[q]; best = q; }                                                           \
        if (best < 0) break; /* minimum(D) == Inf */                                                        \
        int64_t f = best % k, c = best / k;                                                                 \
        labels[f + t * k] = (int32_t)(c + 1);                                                               \
        for (int64_t q = 0; q < k; q++) { D[f + q * k] = (T)INFINITY; D[q + c * k] = (T)INFINITY; }         \
        for (int64_t j = 0; j < len; j++) cent[c * len + j] += Wt[f * len + j];                             \
      }                                                                                                     \
    }                                                                                                       \
    /* Clus:487-496 repairs */                                                                              \
    for (int64_t t = 0; t < R; t++)                                                                         \
      for (int64_t a = 0; a < k; a++)                                                                       \
        if (labels[a + t * k] == 0) labels[a + t * k] = (int32_t)(a + 1);                                   \
    for (int64_t c = 0; c < k; c++)                                                                         \
      for (int64_t j = 0; j < m; j++) centroids[c + j * k] = cent[c * len + j] / (T)R; /* Clus:512-516 */   \
    free(V);                                                                                                \
    free(D);                                                                                                \
    return needfix;                                                                                         \
  }
DEF_CLUSTER(nmfk_or_clustersolutions_f32, float, cosine_f32)
DEF_CLUSTER(nmfk_or_clustersolutions_f64, double, cosine_f64)

/* finalize(Wa, Ha, idx, false)  src/NMFkFinalize.jl:36-79 (silhouette part; means/vars below).
 * Hs: R x (k x m) stacked; labels k x R (1-based).  Outputs: D (kR x kR) cosine distances of
 * zerostoepsilon(vcat(Ha...)) rows with NaN->0 (Fin:52-54), point silhouettes k x R (NaN->0, Fin:58),
 * cluster silhouettes k (Fin:66).  Silhouette definition: Clustering.jl (see header). */
#define DEF_FINALIZE(NAME, T, COS, EPS)                                                                     \
  EXPORT void NAME(const T *Hs, const int32_t *labels, int64_t R, int64_t k, int64_t m, T *D, T *psil,     \
                   T *csil) {                                                                               \
    int64_t nT = k * R;                                                                                     \
    T e2 = (T)(EPS) * (T)(EPS);                                                                             \
    T *Z = (T *)malloc(sizeof(T) * nT * m); /* row p = signal (a of solution r), p = a + r*k */            \
    for (int64_t r = 0; r < R; r++)                                                                         \
      for (int64_t a = 0; a < k; a++)                                                                       \
        for (int64_t j = 0; j < m; j++) {                                                                   \
          T v = Hs[r * k * m + a + j * k];                                                                  \
          Z[(a + r * k) * m + j] = (v < e2) ? e2 : v; /* zerostoepsilon: Help:535-543 */                    \
        }                                                                                                   \
    for (int64_t p = 0; p < nT; p++)                                                                        \
      for (int64_t q = 0; q < nT; q++) {                                                                    \
        T d = (p == q) ? (T)0 : COS(Z + p * m, 1, Z + q * m, 1, m);                                         \
        D[p + q * nT] = (d != d) ? (T)0 : d;                                                                \
      }                                                                                                     \
    int64_t *cnt = (int64_t *)calloc(k, sizeof(int64_t));                                                   \
    for (int64_t p = 0; p < nT; p++) cnt[labels[p] - 1]++;                                                  \
    T *sumd = (T *)malloc(sizeof(T) * k);                                                                   \
    for (int64_t p = 0; p < nT; p++) {                                                                      \
      for (int64_t c = 0; c < k; c++) sumd[c] = 0;                                                          \
      for (int64_t q = 0; q < nT; q++) sumd[labels[q] - 1] += D[p + q * nT];                                \
      int64_t own = labels[p] - 1;                                                                          \
      T s;                                                                                                  \
      if (cnt[own] == 1) {                                                                                  \
        s = 0;                                                                                              \
      } else {                                                                                              \
        T a = sumd[own] / (T)(cnt[own] - 1);                                                                \
        T b = (T)INFINITY;                                                                                  \
        for (int64_t c = 0; c < k; c++)                                                                     \
          if (c != own && cnt[c] > 0) {                                                                     \
            T v = sumd[c] / (T)cnt[c];                                                                      \
            if (v < b) b = v;                                                                               \
          }                                                                                                 \
        s = (a < b) ? (T)1 - a / b : ((a > b) ? b / a - (T)1 : (T)0);                                       \
      }                                                                                                     \
      psil[p] = (s != s) ? (T)0 : s;                                                                        \
    }                                                                                                       \
    for (int64_t c = 0; c < k; c++) {                                                                       \
      T s = 0;                                                                                              \
      int64_t nc = 0;                                                                                       \
      for (int64_t p = 0; p < nT; p++)                                                                      \
        if (labels[p] - 1 == c) { s += psil[p]; nc++; }                                                     \
      csil[c] = s / (T)nc;                                                                                  \
    }                                                                                                       \
    free(Z); free(cnt); free(sumd);                                                                         \
  }
DEF_FINALIZE(nmfk_or_finalize_f32, float, cosine_f32, 1.1920929e-07f)
DEF_FINALIZE(nmfk_or_finalize_f64, double, cosine_f64, 2.220446049250313e-16)

/* cluster means / variances  Fin:64-77 (Statistics.mean, Statistics.var corrected).
 * Ws: R x (n x k), Hs: R x (k x m), labels k x R. */
EXPORT void nmfk_or_cluster_stats(const double *Ws, const double *Hs, const int32_t *labels, int64_t R, int64_t n,
                                  int64_t k, int64_t m, double *Wm, double *Hm, double *Wv, double *Hv) {
  for (int64_t c = 0; c < k; c++) {
    for (int64_t i = 0; i < n; i++) {
      double s = 0, s2 = 0;
      for (int64_t r = 0; r < R; r++)
        for (int64_t a = 0; a < k; a++)
          if (labels[a + r * k] - 1 == c) s += Ws[r * n * k + i + a * n];
      double mean = s / (double)R;
      for (int64_t r = 0; r < R; r++)
        for (int64_t a = 0; a < k; a++)
          if (labels[a + r * k] - 1 == c) { double d = Ws[r * n * k + i + a * n] - mean; s2 += d * d; }
      Wm[i + c * n] = mean;
      Wv[i + c * n] = s2 / (double)(R - 1);
    }
    for (int64_t j = 0; j < m; j++) {
      double s = 0, s2 = 0;
      for (int64_t r = 0; r < R; r++)
        for (int64_t a = 0; a < k; a++)
          if (labels[a + r * k] - 1 == c) s += Hs[r * k * m + a + j * k];
      double mean = s / (double)R;
      for (int64_t r = 0; r < R; r++)
        for (int64_t a = 0; a < k; a++)
          if (labels[a + r * k] - 1 == c) { double d = Hs[r * k * m + a + j * k] - mean; s2 += d * d; }
      Hm[c + j * k] = mean;
      Hv[c + j * k] = s2 / (double)(R - 1);
    }
  }
}

/* ------------------------------------------------------------------------------------------------------
 * robustkmeans(X, k, repeats)  src/NMFkCluster.jl:172-246  
 *
 * The arithmetic lives in Clustering.jl (compat 0.14-0.15, Project.toml:55; un-vendored): kmeans(X, k; maxiter, tol,
 * distance=CosineDist()) = k-means++ seeding (initseeds(:kmpp, X, k): squared Euclidean, the `distance` keyword is
 * not forwarded to the seeding) + Lloyd iterations of _kmeans!: update_centers! (running sum of the member columns in
 * column order, then / count; only clusters whose membership changed), repick_unused_centers (empty clusters: a
 * column drawn with probability ~ its cost, k-means++ like), pairwise(distance, centers, X), update_assignments!
 * (first minimum), objective = sum(costs), converged when |change| < tol.  Restated from the published algorithm;
 * PARITY UNPINNED beyond the reference's own test (test/test_cluster_unit.jl:6-18: valid assignments), because
 * Julia's RNG stream cannot be reproduced: the draws here come from the repo's counter-based generator
 * (u(seed + repeat, draw index)), wsample = StatsBase.sample(Weights): t = u * sum(w), first i with cumsum >= t.
 * robustkmeans keeps the repeat with the lowest total cost (first wins ties, Clus:227) and relabels the clusters by
 * decreasing size (sortclustering, Clus:264-292).  Distances, centres and costs in T; cumulative sums and the
 * objective in double.
 * X: d x n column-major (columns = samples).
 * ------------------------------------------------------------------------------------------------------ */
#define DEF_KMEANS(SUF, T, SQRT)                                                                                        \
  static inline T sqeuclid_##SUF(const T *a, const T *b, int d) {                                                       \
    T s = 0;                                                                                                            \
    for (int i = 0; i < d; i++) {                                                                                       \
      T v = a[i] - b[i];                                                                                                \
      s += v * v;                                                                                                       \
    }                                                                                                                   \
    return s;                                                                                                           \
  }                                                                                                                     \
  static int wsample_##SUF(const T *w, int n, double u) {                                                               \
    double tot = 0;                                                                                                     \
    for (int i = 0; i < n; i++) tot += (double)w[i];                                                                    \
    const double t = u * tot;                                                                                           \
    int i = 0;                                                                                                          \
    double cw = (double)w[0];                                                                                           \
    while (cw < t && i < n - 1) {                                                                                       \
      i++;                                                                                                              \
      cw += (double)w[i];                                                                                               \
    }                                                                                                                   \
    return i;                                                                                                           \
  }                                                                                                                     \
  /* one k-means run; returns the number of iterations */                                                               \
  EXPORT int nmfk_or_kmeans_##SUF(const T *X, int d, int n, int k, int maxiter, double tol, uint64_t seed,             \
                                  int32_t *assign, T *centers, T *costs, int32_t *counts, double *totalcost,            \
                                  int32_t *converged_out) {                                                             \
    uint64_t draw = 0;                                                                                                  \
    T *mc = (T *)malloc(sizeof(T) * (size_t)n), *tc = (T *)malloc(sizeof(T) * (size_t)n);                               \
    uint8_t *upd = (uint8_t *)malloc((size_t)k);                                                                        \
    int32_t *unused = (int32_t *)malloc(sizeof(int32_t) * (size_t)k), *wc = (int32_t *)malloc(sizeof(int32_t) * (size_t)k); \
    int nun = 0;                                                                                                        \
    /* k-means++ seeding, squared Euclidean */                                                                          \
    int p = (int)(nmfk_uniform(seed, draw++) * (double)n);                                                              \
    if (p > n - 1) p = n - 1;                                                                                           \
    for (int i = 0; i < d; i++) centers[i] = X[i + (size_t)p * d];                                                      \
    if (k > 1) {                                                                                                        \
      for (int j = 0; j < n; j++) mc[j] = sqeuclid_##SUF(X + (size_t)j * d, X + (size_t)p * d, d);                      \
      mc[p] = 0;                                                                                                        \
      for (int c = 1; c < k; c++) {                                                                                     \
        p = wsample_##SUF(mc, n, nmfk_uniform(seed, draw++));                                                           \
        for (int i = 0; i < d; i++) centers[i + (size_t)c * d] = X[i + (size_t)p * d];                                  \
        for (int j = 0; j < n; j++) {                                                                                   \
          const T v = sqeuclid_##SUF(X + (size_t)j * d, X + (size_t)p * d, d);                                          \
          if (v < mc[j]) mc[j] = v;                                                                                     \
        }                                                                                                               \
        mc[p] = 0;                                                                                                      \
      }                                                                                                                 \
    }                                                                                                                   \
    int it = 0, converged = 0;                                                                                          \
    double objv = 0, prev = 0;                                                                                          \
    for (int pass = 0;; pass++) { /* pass 0 = initial assignment */                                                     \
      if (pass > 0) {                                                                                                   \
        it++;                                                                                                           \
        for (int c = 0; c < k; c++)                                                                                     \
          if (upd[c]) wc[c] = 0;                                                                                        \
        for (int j = 0; j < n; j++) { /* update_centers! */                                                             \
          const int c = assign[j];                                                                                      \
          if (!upd[c]) continue;                                                                                        \
          if (wc[c] > 0)                                                                                                \
            for (int i = 0; i < d; i++) centers[i + (size_t)c * d] += X[i + (size_t)j * d];                             \
          else                                                                                                          \
            for (int i = 0; i < d; i++) centers[i + (size_t)c * d] = X[i + (size_t)j * d];                              \
          wc[c]++;                                                                                                      \
        }                                                                                                               \
        for (int c = 0; c < k; c++)                                                                                     \
          if (upd[c])                                                                                                   \
            for (int i = 0; i < d; i++) centers[i + (size_t)c * d] /= (T)wc[c];                                         \
        if (nun > 0) { /* repick_unused_centers */                                                                      \
          for (int j = 0; j < n; j++) tc[j] = costs[j];                                                                 \
          for (int q = 0; q < nun; q++) {                                                                               \
            const int c = unused[q];                                                                                    \
            const int j = wsample_##SUF(tc, n, nmfk_uniform(seed, draw++));                                             \
            tc[j] = 0;                                                                                                  \
            for (int i = 0; i < d; i++) centers[i + (size_t)c * d] = X[i + (size_t)j * d];                              \
            for (int jj = 0; jj < n; jj++) {                                                                            \
              const T v = cosine_##SUF(X + (size_t)j * d, 1, X + (size_t)jj * d, 1, d);                                 \
              if (v < tc[jj]) tc[jj] = v;                                                                               \
            }                                                                                                           \
          }                                                                                                             \
        }                                                                                                               \
      }                                                                                                                 \
      /* update_assignments! */                                                                                         \
      for (int c = 0; c < k; c++) {                                                                                     \
        counts[c] = 0;                                                                                                  \
        upd[c] = (pass == 0);                                                                                           \
      }                                                                                                                 \
      nun = 0;                                                                                                          \
      for (int j = 0; j < n; j++) {                                                                                     \
        int a = 0;                                                                                                      \
        T cm = cosine_##SUF(centers, 1, X + (size_t)j * d, 1, d);                                                       \
        for (int c = 1; c < k; c++) {                                                                                   \
          const T ci = cosine_##SUF(centers + (size_t)c * d, 1, X + (size_t)j * d, 1, d);                               \
          if (ci < cm) {                                                                                                \
            a = c;                                                                                                      \
            cm = ci;                                                                                                    \
          }                                                                                                             \
        }                                                                                                               \
        if (pass == 0)                                                                                                  \
          assign[j] = a;                                                                                                \
        else if (assign[j] != a) {                                                                                      \
          upd[a] = 1;                                                                                                   \
          upd[assign[j]] = 1;                                                                                           \
          assign[j] = a;                                                                                                \
        }                                                                                                               \
        costs[j] = cm;                                                                                                  \
        counts[a]++;                                                                                                    \
      }                                                                                                                 \
      for (int c = 0; c < k; c++)                                                                                       \
        if (counts[c] == 0) {                                                                                           \
          unused[nun++] = c;                                                                                            \
          upd[c] = 0;                                                                                                   \
        }                                                                                                               \
      prev = objv;                                                                                                      \
      objv = 0;                                                                                                         \
      for (int j = 0; j < n; j++) objv += (double)costs[j];                                                             \
      if (pass > 0) {                                                                                                   \
        const double ch = objv - prev;                                                                                  \
        if (!(ch > tol) && (k == 1 || fabs(ch) < tol)) converged = 1;                                                   \
      }                                                                                                                 \
      if (converged || it >= maxiter) break;                                                                            \
      /* (the reference's to_update[unused] .= true after the repick only selects which rows of the distance matrix   \
       * are recomputed; all of them are recomputed here, same values) */                                               \
    }                                                                                                                   \
    *totalcost = objv;                                                                                                  \
    *converged_out = converged;                                                                                         \
    free(mc);                                                                                                           \
    free(tc);                                                                                                           \
    free(upd);                                                                                                          \
    free(unused);                                                                                                       \
    free(wc);                                                                                                           \
    return it;                                                                                                          \
  }
DEF_KMEANS(f32, float, sqrtf)
DEF_KMEANS(f64, double, sqrt)

/* sortclustering(c::KmeansResult)  Clus:264-292: clusters relabelled 1..k' by decreasing size (stable: ties keep the
 * order of first appearance).  assign: 0-based in, 1-based out; perm[new] = old cluster index; returns k'. */
EXPORT int nmfk_or_sortclustering(int32_t *assign, int n, const int32_t *counts, int k, int32_t *perm) {
  int32_t *first = (int32_t *)malloc(sizeof(int32_t) * (size_t)k), *seen = (int32_t *)calloc((size_t)k, sizeof(int32_t));
  int nj = 0;
  for (int j = 0; j < n; j++)
    if (!seen[assign[j]]) {
      seen[assign[j]] = 1;
      first[nj++] = assign[j];
    }
  /* stable insertion sort of the appearance list by count, descending */
  for (int a = 1; a < nj; a++) {
    const int32_t v = first[a];
    int b = a - 1;
    while (b >= 0 && counts[first[b]] < counts[v]) {
      first[b + 1] = first[b];
      b--;
    }
    first[b + 1] = v;
  }
  for (int c = 0; c < k; c++) seen[c] = 0;
  for (int a = 0; a < nj; a++) {
    seen[first[a]] = a + 1;
    perm[a] = first[a];
  }
  for (int j = 0; j < n; j++) assign[j] = seen[assign[j]];
  free(first);
  free(seen);
  return nj;
}

/* robustkmeans(X, k, repeats)  Clus:172-246 (without the JLD cache): best of `repeats` runs, sorted.
 * assign (n, 1-based), centers (d x k, sorted order), counts (k, sorted), costs (n).  Returns k' (clusters found). */
#define DEF_ROBUST(SUF, T)                                                                                             \
  EXPORT int nmfk_or_robustkmeans_##SUF(const T *X, int d, int n, int k, int repeats, int maxiter, double tol,         \
                                        uint64_t seed, int32_t *assign, T *centers, T *costs, int32_t *counts,         \
                                        double *totalcost, int32_t *best_repeat, int32_t *iterations,                  \
                                        double *all_costs /* repeats or NULL */) {                                     \
    int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *cn = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);  \
    T *ce = (T *)malloc(sizeof(T) * (size_t)d * k), *co = (T *)malloc(sizeof(T) * (size_t)n);                           \
    double best = 0;                                                                                                   \
    for (int r = 0; r < repeats; r++) {                                                                                \
      double tc;                                                                                                       \
      int32_t conv;                                                                                                    \
      const int it = nmfk_or_kmeans_##SUF(X, d, n, k, maxiter, tol, seed + (uint64_t)r, a, ce, co, cn, &tc, &conv);     \
      if (all_costs) all_costs[r] = tc;                                                                                \
      if (r == 0 || tc < best) { /* Clus:227 */                                                                        \
        best = tc;                                                                                                     \
        *best_repeat = r;                                                                                              \
        *iterations = it;                                                                                              \
        memcpy(assign, a, sizeof(int32_t) * (size_t)n);                                                                \
        memcpy(centers, ce, sizeof(T) * (size_t)d * k);                                                                \
        memcpy(costs, co, sizeof(T) * (size_t)n);                                                                      \
        memcpy(counts, cn, sizeof(int32_t) * (size_t)k);                                                               \
      }                                                                                                                \
    }                                                                                                                  \
    *totalcost = best;                                                                                                 \
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);                                                    \
    const int kf = nmfk_or_sortclustering(assign, n, counts, k, perm);                                                 \
    memcpy(ce, centers, sizeof(T) * (size_t)d * k);                                                                    \
    memcpy(cn, counts, sizeof(int32_t) * (size_t)k);                                                                   \
    for (int c = 0; c < k; c++) {                                                                                      \
      counts[c] = c < kf ? cn[perm[c]] : 0;                                                                            \
      for (int i = 0; i < d; i++) centers[i + (size_t)c * d] = c < kf ? ce[i + (size_t)perm[c] * d] : (T)0;            \
    }                                                                                                                  \
    free(a);                                                                                                           \
    free(cn);                                                                                                          \
    free(ce);                                                                                                          \
    free(co);                                                                                                          \
    free(perm);                                                                                                        \
    return kf;                                                                                                         \
  }
DEF_ROBUST(f32, float)
DEF_ROBUST(f64, double)

/* Clustering.silhouettes(assignments, counts, dists) with dists = pairwise(CosineDist(), zerostoepsilon(X); dims=2)
 * (Clus:204-213).  assign 1-based; e2 = eps(T)^2.  Point i: a = mean distance to the OTHER members of its cluster,
 * b = min over other clusters of the mean distance; s = (b - a) / max(a, b); singleton cluster: 0. */
#define DEF_SIL(SUF, T, E2)                                                                                            \
  EXPORT void nmfk_or_point_silhouettes_##SUF(const T *X, int d, int n, const int32_t *assign, int k, T *sil) {         \
    T *Z = (T *)malloc(sizeof(T) * (size_t)d * n);                                                                     \
    for (size_t i = 0; i < (size_t)d * n; i++) Z[i] = X[i] < (T)(E2) ? (T)(E2) : X[i];                                 \
    int32_t *cnt = (int32_t *)calloc((size_t)k, sizeof(int32_t));                                                      \
    for (int j = 0; j < n; j++) cnt[assign[j] - 1]++;                                                                  \
    T *s = (T *)malloc(sizeof(T) * (size_t)k);                                                                         \
    for (int i = 0; i < n; i++) {                                                                                      \
      for (int c = 0; c < k; c++) s[c] = 0;                                                                            \
      for (int j = 0; j < n; j++)                                                                                      \
        if (j != i) s[assign[j] - 1] += cosine_##SUF(Z + (size_t)i * d, 1, Z + (size_t)j * d, 1, d);                   \
      const int ci = assign[i] - 1;                                                                                    \
      if (cnt[ci] <= 1) {                                                                                              \
        sil[i] = 0;                                                                                                    \
        continue;                                                                                                      \
      }                                                                                                                \
      const T a = s[ci] / (T)(cnt[ci] - 1);                                                                            \
      T b = 0;                                                                                                         \
      int have = 0;                                                                                                    \
      for (int c = 0; c < k; c++) {                                                                                    \
        if (c == ci || cnt[c] == 0) continue;                                                                          \
        const T v = s[c] / (T)cnt[c];                                                                                  \
        if (!have || v < b) b = v;                                                                                     \
        have = 1;                                                                                                      \
      }                                                                                                                \
      const T mx = a > b ? a : b;                                                                                      \
      sil[i] = have ? (b - a) / mx : 0;                                                                                \
    }                                                                                                                  \
    free(Z);                                                                                                           \
    free(cnt);                                                                                                         \
    free(s);                                                                                                           \
  }
DEF_SIL(f32, float, 1.4210854715202004e-14)
DEF_SIL(f64, double, 4.930380657631324e-32)
