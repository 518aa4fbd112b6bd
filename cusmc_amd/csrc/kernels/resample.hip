// Metropolis resampler on gfx950.
//
// Replaces the inner loop of Sampler::metropolis_hastings (src/samplers.cpp:21-35), which the
// reference runs on the CPU even in its GPU build, with a shared (racy) u/j/k and a shared
// mt19937 under `omp parallel for`.  One lane = one chain i:
//     k = i;  B times { u ~ U[0,1); j ~ UnifInt[0,N); if (u <= w[j] / w[k]) k = j; }  a[i] = k
// The draw order (u, then j), the division and the `<=` are the reference's, so a NaN ratio never
// accepts (man/metropolis_hastings.Rd:22-27: w = c(0,0) gives a = c(0,1)).  fp64 division on
// gfx950 is correctly rounded (no fast-math here), so with the shared Philox contract
// (philox.h) the index sequence is bit-identical to oracle_metropolis().
//
// Cost model: per step one Philox4x32-10 block (~70 integer VALU ops), one 8-byte
// random gather of w[j] -- w is 0.8 MB (N = 1e5, L2-resident) or 8 MB (N = 1e6,
// Infinity-Cache-resident) -- and one fp64 divide.  The chain itself is metropolis_chain()
// in smallops.h (shared with the fused filter step).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "smallops.h"

namespace cusmc {

__global__ __launch_bounds__(256) void metropolis_kernel(const double *__restrict__ w, uint32_t N,
                                                         uint32_t B, uint32_t k0, uint32_t k1,
                                                         uint32_t step, uint32_t first,
                                                         uint32_t count, uint32_t *__restrict__ a)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
    a[t] = metropolis_chain(w, N, B, first + t, step, k0, k1);
  }
}

__global__ __launch_bounds__(256) void metropolis_hi_kernel(const double *__restrict__ w,
                                                            const uint32_t *__restrict__ whi, uint32_t N,
                                                            uint32_t B, uint32_t k0, uint32_t k1,
                                                            uint32_t step, uint32_t first,
                                                            uint32_t count, uint32_t *__restrict__ a)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
    a[t] = metropolis_chain_hi(w, whi, N, B, first + t, step, k0, k1);
  }
}

__global__ __launch_bounds__(256) void metropolis_log_kernel(const double *__restrict__ lw, uint32_t N,
                                                             uint32_t B, uint32_t k0, uint32_t k1,
                                                             uint32_t step, uint32_t first,
                                                             uint32_t count, uint32_t *__restrict__ a)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
    a[t] = metropolis_chain_log(lw, N, B, first + t, step, k0, k1);
  }
}

hipError_t launch_metropolis_log(const double *lw, uint32_t N, uint32_t B, uint64_t seed, uint32_t step,
                                 uint32_t first, uint32_t count, uint32_t *a, int num_cus, hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  long blocks = ((long)count + 255) / 256;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(metropolis_log_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, lw, N, B,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, first, count, a);
  return hipGetLastError();
}

// whi[i] = high word of w[i]
__global__ __launch_bounds__(256) void hiword_kernel(const double *__restrict__ w, uint32_t N,
                                                     uint32_t *__restrict__ whi)
{
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < N; i += gridDim.x * 256u)
    whi[i] = (uint32_t)(__builtin_bit_cast(uint64_t, w[i]) >> 32);
}

// The truncated-table chain with the head of the table in LDS (r03).  The resampler is bound by the rate of its
// random gathers -- one L2 request per lane and step, ~0.45 per cycle and CU, whatever else the wave does
// (profiles/r03_pmc_mh.md) -- so every gather served from LDS instead is time saved: one workgroup per CU keeps the
// first L = min(N, 40 000) high words (160 KB) and reads the rest from L2 as before.  BASELINE configs[1] (N = 1e5):
// 40 % of the gathers.  Same words, same index sequence.
__global__ __launch_bounds__(1024) void metropolis_hi_lds_kernel(const double *__restrict__ w,
                                                                 const uint32_t *__restrict__ whi, uint32_t N, uint32_t L,
                                                                 uint32_t B, uint32_t k0, uint32_t k1, uint32_t step,
                                                                 uint32_t first, uint32_t count, uint32_t *__restrict__ a)
{
  extern __shared__ uint32_t lds_tab[];
  {
    // fill: 16 bytes per lane and load, four loads in flight (L is a multiple of 4; whi comes from hipMalloc)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 *src = reinterpret_cast<const u4 *>(whi);
    u4 *dst = reinterpret_cast<u4 *>(lds_tab);
    const uint32_t L4 = L >> 2;
    for (uint32_t t0 = threadIdx.x; t0 < L4; t0 += 4 * blockDim.x) {
      u4 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t t = t0 + c * blockDim.x;
        if (t < L4) v[c] = src[t];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t t = t0 + c * blockDim.x;
        if (t < L4) dst[t] = v[c];
      }
    }
  }
  __syncthreads();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
    a[t] = metropolis_chain_hi(w, whi, N, B, first + t, step, k0, k1, lds_tab, L);
  }
}

// The truncated-table chain pays off once the doubles no longer fit one XCD's L2 (4 MB) -- or, with the head of the
// table in LDS, as soon as there are enough steps per chain to pay for filling it.
bool metropolis_wants_hiwords(uint32_t N)
{
  if (const char *e = getenv("CUSMC_MH_HI")) return e[0] == '1';  // (switch: for A/B timing)
  return (uint64_t)N * 8 > (3u << 20);
}
bool metropolis_wants_lds_table(uint32_t N, uint32_t B, uint32_t count)
{
  if (const char *e = getenv("CUSMC_MH_LDS")) return e[0] == '1';  // (switch: for A/B timing)
  // Measured (scripts/calib/mh_occupancy.py, B = 1000, us without -> with): N = 32768: 200 -> 268, 65536: 257 -> 294,
  // 1e5: 409 -> 344, 131072: 489 -> 367, 196608: 740 -> 590, 262144: 993 -> 868.  It pays from ~1.5 waves per SIMD up
  // (below that a wave's own latency is what counts, and the branchy two-source gather has more of it); the fill is
  // 160 KB per CU from L2, a few microseconds: worth it from ~100 steps per chain; one workgroup of <= 1024 chains per
  // CU covers 262144 chains in one pass; above 4e5 weights less than a tenth of the gathers would land in LDS.
  return B >= 96 && N <= 400000 && count >= 90000 && count <= 262144;
}

hipError_t launch_hiwords(const double *w, uint32_t N, uint32_t *whi, int num_cus, hipStream_t stream)
{
  if (N == 0) return hipSuccess;
  long blocks = ((long)N + 255) / 256;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(hiword_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, N, whi);
  return hipGetLastError();
}

// whi == NULL: gathers from the doubles themselves.
hipError_t launch_metropolis(const double *w, const uint32_t *whi, uint32_t N, uint32_t B, uint64_t seed,
                             uint32_t step, uint32_t first, uint32_t count, uint32_t *a,
                             int num_cus, hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  if (whi && metropolis_wants_lds_table(N, B, count)) {
    const uint32_t L = (N < 40000u ? N : 40000u) & ~3u;  // 160 000 of the CU's 163 840 bytes
    const size_t lds_bytes = (size_t)L * 4;
    static LdsConfig lds_configured;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(metropolis_hi_lds_kernel), lds_bytes, lds_configured); e != hipSuccess) return e;
    long blocks = num_cus;  // one workgroup per CU (LDS), all of the CU's chains in it
    if (blocks > ((long)count + 63) / 64) blocks = ((long)count + 63) / 64;
    long threads = (((long)count + blocks - 1) / blocks + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(metropolis_hi_lds_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), lds_bytes, stream, w, whi,
                       N, L, B, (uint32_t)seed, (uint32_t)(seed >> 32), step, first, count, a);
    return hipGetLastError();
  }
  if (whi) {
    long blocks = ((long)count + 255) / 256;
    const long cap = (long)num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(metropolis_hi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, whi, N, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), step, first, count, a);
    return hipGetLastError();
  }
  long blocks = ((long)count + 255) / 256;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(metropolis_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, N, B,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, first, count, a);
  return hipGetLastError();
}

}  // namespace cusmc
