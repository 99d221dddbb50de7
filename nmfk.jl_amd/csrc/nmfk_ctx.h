// Private to libnmfk_hip: the context object behind the C ABI, shared by nmfk_api.hip and nmfk_comm.hip.
#pragma once
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/nmfk_hip.h"
#include "nmfk_common.h"

#define NMFK_EXPORT extern "C" __attribute__((visibility("default")))

// message of the last failure on the calling thread (nmfk_last_error)
inline std::string &nmfk_error_slot() {
  static thread_local std::string s;
  return s;
}
inline int fail(int code, const std::string &msg) {
  nmfk_error_slot() = msg;
  return code;
}

#define HIPCHECK(expr)                                                                              \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      char _b[512];                                                                                 \
      snprintf(_b, sizeof(_b), "HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__, \
               #expr);                                                                              \
      return fail(NMFK_ERR_HIP, _b);                                                                \
    }                                                                                               \
  } while (0)

struct DevBuf {
  char *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + (bytes >> 3) + 4096;
    if (hipMalloc((void **)&p, want) != hipSuccess) return 1;
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

enum { PK_HSTEP = 0, PK_WSTEP = 1 };

struct nmfk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<hipStream_t> gstreams;  // one per concurrently running rank group of a sweep
  hipStream_t poll_stream = nullptr;  // copies the unit states back without blocking the compute streams
  hipDeviceProp_t prop;
  // data
  int64_t n = 0, m = 0;
  float *Xc = nullptr, *Xr = nullptr;
  float *Wgt = nullptr;  // optional n x m weight array of the monitored objective
  // sparse X (nmfk_set_X_csc): CSC for the H half-step, CSR for the W half-step and the objective
  bool sparse = false;
  int64_t nnz = 0;
  // the non-zeros are (index, value) records of 8 bytes, so that a walk costs one load per non-zero
  int32_t *colptr = nullptr, *rowptr = nullptr;
  // blocked form of the sparse half-steps: sliced ELL of the rows ([0]: W half-step) and of the columns ([1]: H half-step),
  // see NmfkSparseArgs::ell; null when the padding would exceed NMFK_ELL_MAX_PAD x the non-zeros (skewed lane elements)
  int2 *ell[2] = {nullptr, nullptr};
  int32_t *ellptr[2] = {nullptr, nullptr};
  int32_t ell_ngb[2] = {0, 0};
  double ell_pad[2] = {0, 0};  // slots / non-zeros
  int2 *rec_csc = nullptr, *rec_csr = nullptr;
  int64_t nan_count = 0, zero_count = 0;
  double lambda = 1e-32;
  // workspaces
  DevBuf arena;    // sweep
  DevBuf scratch;  // set_X staging, clustering
  DevBuf xtile;    // tiled copies of X for the split-operand MFMA half-step (built on first use after set_X)
  uint64_t xgen = 0, xtile_gen = ~(uint64_t)0;
  void *pinned = nullptr;
  size_t pinned_cap = 0;
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> events;
  struct ProfEntry {
    double ms = 0, flops = 0;
    int64_t launches = 0;
  };
  std::map<std::string, ProfEntry> prof;
  int32_t sweep_info[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // nmfk_last_sweep_info (8) / nmfk_last_sweep_info_ex (16)
  // nmfk_set_objective_trace: the monitored objective (Mult:74) of every unit at every check of the last sweep
  bool trace_objective = false;
  std::vector<double> obj_trace;  // [unit][check]
  std::vector<int32_t> obj_trace_unit;  // (kidx * nruns + restart) -> unit
  int32_t obj_trace_stride = 0, obj_trace_nruns = 0;
};

