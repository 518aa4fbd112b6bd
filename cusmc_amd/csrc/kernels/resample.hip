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

// The truncated-table chain pays off once the doubles no longer fit one XCD's L2 (4 MB).
bool metropolis_wants_hiwords(uint32_t N)
{
  if (const char *e = getenv("CUSMC_MH_HI")) return e[0] == '1';  // (switch: for A/B timing)
  return (uint64_t)N * 8 > (3u << 20);
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
