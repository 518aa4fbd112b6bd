// Ancestor rows of a SHARDED particle array: x_anc[i] = x_{t-1}[a[i]] where row j of x_{t-1} lives on the
// device that owns particle j.  The multi-device filter (cusmc_pf_run_multi_host) calls this once per time
// step between the resampler and the proposal: the reference's propagate_K reads post_x_t[t-1][a_t[t*N+i]]
// from one host array (src/mcmc.cpp:118-121); here the rows come straight out of the peers' HBM through
// the pointers of `tab` (peer access over xGMI, or plain local pointers when several shards share a device).
// Only the N/R rows this shard's ancestors name cross the fabric -- not the whole of x_{t-1}.
#include <hip/hip_runtime.h>

#include "../launch.h"

namespace cusmc {

// one thread per element: consecutive lanes read consecutive doubles of one row (512 contiguous bytes per
// wave at d = 64), neighbouring rows are unrelated anyway.  d and the shard bounds are wave-uniform.
__global__ __launch_bounds__(256) void gather_rows_sharded_kernel(ShardTable tab, const uint32_t *__restrict__ a,
                                                                  uint32_t count, int d, double *__restrict__ out)
{
  const size_t total = (size_t)count * d;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const uint32_t i = (uint32_t)(idx / (unsigned)d);
    const int k = (int)(idx - (size_t)i * d);
    const uint32_t j = a[i];
    int r = 0;
    while (r + 1 < tab.n && j >= tab.first[r + 1]) ++r;
    out[idx] = tab.base[r][(size_t)(j - tab.first[r]) * d + k];
  }
}

hipError_t launch_gather_rows_sharded(const ShardTable &tab, const uint32_t *a, uint32_t count, int d, double *out,
                                      int num_cus, hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  const size_t total = (size_t)count * d;
  size_t blocks = (total + 255) / 256;
  const size_t cap = (size_t)num_cus * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(gather_rows_sharded_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tab, a, count, d, out);
  return hipGetLastError();
}

}  // namespace cusmc
