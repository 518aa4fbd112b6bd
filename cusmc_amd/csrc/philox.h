// Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw: "Parallel random numbers:
// as easy as 1, 2, 3", SC'11) and the derived draws of the build's RNG contract.
//
// The reference cannot be seeded at all (src/samplers.cpp:10-11, src/statistics.cc.cpp:231-232,
// src/mvn_dist.cu.cpp:187-189), and its device path keeps a 48-byte curandState per ELEMENT
// (mvn_sample_setup_kernel, src/mvn_dist.cu.cpp:15-22: 3 GB at 1e6 x 64).  Here every draw is a
// pure function of (seed, index, sub, step, domain): no state array, reproducible, shardable.
//
//   key     = (seed_lo, seed_hi)
//   counter = (index, sub, step, domain)
//   domain  : 1 resampler          sub = iteration n          words 0,1 -> u; 2,3 -> j
//             2 proposal normals   sub = component pair j/2   Box-Muller on (0,1],(0,1)
//             3 chi-square normals sub = j*64 + attempt
//             4 initial normals    sub = component pair
//             5 chi-square accept / boost uniforms
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cusmc {

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                        uint32_t c3, uint32_t k0, uint32_t k1)
{
#ifdef CUSMC_ABL_NO_PHILOX  // ablation builds only (scripts/calib/prop_time.py): what the ten rounds cost a kernel
  return u32x4{c0 * 2654435761u + k0, c1 ^ c0, c2 + k1, c3 ^ c1 * 40503u};
#endif
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

// 53-bit uniform in [0,1).
__host__ __device__ __forceinline__ double u01_53(uint32_t hi, uint32_t lo)
{
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * 0x1.0p-53;
}

// floor(v * N / 2^64), v the 64-bit word (hi,lo): uniform on [0,N) to within N * 2^-64.
__host__ __device__ __forceinline__ uint32_t uint_below(uint32_t hi, uint32_t lo, uint32_t N)
{
  // (hi*2^32 + lo) * N >> 64  ==  (hi*N + ((lo*N) >> 32)) >> 32
  const uint64_t t = (uint64_t)hi * N + (((uint64_t)lo * N) >> 32);
  return (uint32_t)(t >> 32);
}

}  // namespace cusmc
