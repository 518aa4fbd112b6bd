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
//   domain  : 1 resampler          sub = iteration n / 2      half n % 2: word 2h -> leading bits of u, 2h+1 -> j
//             7 resampler: the next 53 bits of u (sub = n), when the first 32 cannot decide
//             16+q resampler: index redraws (sub = n), Lemire's rejection
//             2 proposal normals   sub = component pair j/2   Box-Muller on (0,1],(0,1)
//             3 chi-square normals sub = j*64 + attempt
//             4 initial normals    sub = component pair
//             5 chi-square accept / boost uniforms
//             6 chi-square closed form (integer nu <= 16): uniforms, sub = pair + (block << 16)
//             8 chi-square closed form, odd nu: the pair's two normals, sub = pair
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

// ---- resampler draws, RNG contract 3 (restated in oracle/cusmc_oracle.c: mh_index, mh_accept) ----------------------
// One block (i, n / 2, step, 1) serves chain steps n and n + 1: half h = n % 2 gives a = word 2h (the leading 32 bits
// of u) and b = word 2h + 1 (the index candidate).  Both draws stay exact: more bits are read only when these cannot
// decide (probability 2^-32 and < N 2^-32 per step).

// tN = (2^32 - N) mod N: candidates whose low product word is below it are redrawn (Lemire's unbiased bounded integer)
__host__ __device__ __forceinline__ uint32_t mh_tn(uint32_t N) { return (0u - N) % N; }

// j uniform on [0, N) from the candidate b; redraws come from blocks (i, n, step, 16 + q), words 0..3 in order
__host__ __device__ __forceinline__ uint32_t mh_index(uint32_t b, uint32_t i, uint32_t n, uint32_t step, uint32_t N,
                                                      uint32_t tN, uint32_t k0, uint32_t k1)
{
  uint64_t m = (uint64_t)b * N;
  if ((uint32_t)m < tN) {
    for (uint32_t q = 0;; ++q) {
      const u32x4 r = philox4x32_10(i, n, step, 16u + q, k0, k1);
      m = (uint64_t)r.x * N;
      if ((uint32_t)m >= tN) break;
      m = (uint64_t)r.y * N;
      if ((uint32_t)m >= tN) break;
      m = (uint64_t)r.z * N;
      if ((uint32_t)m >= tN) break;
      m = (uint64_t)r.w * N;
      if ((uint32_t)m >= tN) break;
    }
  }
  return (uint32_t)(m >> 32);
}

// the reference's test u <= r (src/samplers.cpp:29) for the uniform real u whose leading 32 bits are a
__host__ __device__ __forceinline__ bool mh_accept(uint32_t a, double r, uint32_t i, uint32_t n, uint32_t step,
                                                   uint32_t k0, uint32_t k1)
{
  const double lo = (double)a * 0x1.0p-32, hi = lo + 0x1.0p-32;  // both exact
  bool acc = hi <= r;                                             // (false for a NaN ratio: never accepts)
  if (!acc && lo <= r) {                                          // r inside u's cell: the next 53 bits decide
    const u32x4 x = philox4x32_10(i, n, step, 7u, k0, k1);
    acc = u01_53(x.x, x.y) <= r * 0x1.0p32 - (double)a;           // r 2^32 in [a, a + 1]: the difference is exact
  }
  return acc;
}

}  // namespace cusmc
