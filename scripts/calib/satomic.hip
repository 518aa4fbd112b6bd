// Does gfx950 execute SCALAR memory atomics (s_atomic_add ... glc), are they coherent across the 8
// XCDs, and what does one cost?  They return through lgkmcnt, so a wave with vector loads in
// flight can take a ticket without draining vmcnt -- which is what a work queue wants.
// Not product code.  Build: hipcc --offload-arch=gfx950 -O2 -o scripts/calib/satomic scripts/calib/satomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void tickets_scalar(unsigned *ctr, unsigned *hit, unsigned long long *lat, int iters)
{
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) hit[v] += 1;  // tickets are unique, so this is race-free iff the atomic works
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) lat[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void tickets_vector(unsigned *ctr, unsigned *hit, unsigned long long *lat, int iters)
{
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    unsigned v = 0;
    if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __builtin_amdgcn_readfirstlane(v);
    if ((threadIdx.x & 63) == 0) hit[v] += 1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) lat[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main()
{
  const int blocks = 256, threads = 512, waves = blocks * threads / 64;
  for (int iters : {1, 4, 16}) {
    for (int kind = 0; kind < 2; ++kind) {
      const int total = waves * iters;
      unsigned *ctr, *hit; unsigned long long *lat;
      if (hipMalloc(&ctr, 64) != hipSuccess || hipMalloc(&hit, total * 4) != hipSuccess || hipMalloc(&lat, waves * 8) != hipSuccess) return 1;
      (void)hipMemset(ctr, 0, 64); (void)hipMemset(hit, 0, total * 4);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(tickets_scalar, dim3(blocks), dim3(threads), 0, 0, ctr, hit, lat, iters);
      else hipLaunchKernelGGL(tickets_vector, dim3(blocks), dim3(threads), 0, 0, ctr, hit, lat, iters);
      (void)hipEventRecord(e1);
      hipError_t err = hipDeviceSynchronize();
      if (err != hipSuccess) { printf("%s atomics: kernel failed: %s\n", kind ? "vector" : "scalar", hipGetErrorString(err)); return 2; }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned> h(total); unsigned c = 0; std::vector<unsigned long long> l(waves);
      (void)hipMemcpy(h.data(), hit, total * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(l.data(), lat, waves * 8, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < total; ++i) bad += h[i] != 1;
      double lm = 0; for (auto v : l) lm += (double)v / waves;
      printf("%s atomics, %d waves x %d: counter %u (want %d), tickets not hit exactly once: %d, kernel %.1f us = %.0f atomics/us, %.0f memtime ticks per atomic per wave\n",
             kind ? "vector" : "scalar", waves, iters, c, total, bad, ms * 1e3, total / (ms * 1e3), lm / iters);
      (void)hipFree(ctr); (void)hipFree(hit); (void)hipFree(lat);
    }
  }
  return 0;
}
