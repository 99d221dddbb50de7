// Counter-based U(0,1) generator shared bit-for-bit by the HIP library and the CPU oracle
// (oracle/nmfk_oracle.c: nmfk_uniform).  It stands in for Julia's `rand(n,k)` / `rand(k,m)`
// (src/NMFkMultiplicative.jl:38,48), whose stream cannot be reproduced outside Julia: a restart is
// identified by its seed, element i of the "W then H" draw order is u(seed, i).
// Values are odd 24-bit integers * 2^-24: never 0 or 1, exactly representable in fp32 and fp64.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NMFK_HD __host__ __device__ __forceinline__
#else
#define NMFK_HD static inline
#endif

NMFK_HD uint64_t nmfk_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// key = nmfk_splitmix64(seed) can be hoisted out of loops
NMFK_HD float nmfk_uniform_keyed(uint64_t key, uint64_t idx) {
  uint64_t z = nmfk_splitmix64(key ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
  uint32_t b = (uint32_t)(z >> 41);  // 23 bits
  return (float)(2u * b + 1u) * (1.0f / 16777216.0f);
}
