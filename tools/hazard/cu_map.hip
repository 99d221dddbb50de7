// Which physical CUs does a process reach under HSA_CU_MASK?  One wave per workgroup records XCC_ID and HW_ID.
// hipcc --offload-arch=gfx950 -O2 tools/hazard/cu_map.hip -o tools/hazard/cu_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned *out) {
  // spin a little so that the workgroups spread over every CU the queue may use
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 20000) {
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(4 | (31 << 11));  // HW_REG_HW_ID
  }
}
int main() {
  const int nwg = 16384;
  unsigned *d;
  hipMalloc((void **)&d, sizeof(unsigned) * 2 * nwg);
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(64), 0, 0, d);
  std::vector<unsigned> h(2 * nwg);
  hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> cus;  // xcc -> {(se, sh, cu)}
  for (int i = 0; i < nwg; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    cus[xcc].insert(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15));
  }
  int total = 0;
  for (auto &kv : cus) {
    printf("xcc %u: %zu CUs:", kv.first, kv.second.size());
    for (unsigned c : kv.second) printf(" %x", c);
    printf("\n");
    total += (int)kv.second.size();
  }
  printf("total CUs seen: %d\n", total);
  return 0;
}
