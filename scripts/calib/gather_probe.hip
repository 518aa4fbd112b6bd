// How much faster are the resampler's random weight gathers when the table fits one XCD's 4 MB L2?
// 1e6 chains x 10 Philox-indexed gathers (the C3 resample step) from: N doubles (8 MB at N = 1e6),
// N 32-bit words (4 MB), N 16-bit words (2 MB).  Not product code.
// Build: hipcc --offload-arch=gfx950 -O3 -I cusmc_amd/csrc -o scripts/calib/gather_probe scripts/calib/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "philox.h"
using namespace cusmc;

template <class T>
__global__ __launch_bounds__(256) void chains(const T *__restrict__ w, uint32_t N, uint32_t B, double *out)
{
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < N; i += gridDim.x * 256u) {
    double acc = 0;
    for (uint32_t n = 0; n < B; n += 2) {
      const u32x4 r0 = philox4x32_10(i, n, 1u, 1u, 7u, 9u), r1 = philox4x32_10(i, n + 1, 1u, 1u, 7u, 9u);
      const uint32_t j0 = uint_below(r0.z, r0.w, N), j1 = uint_below(r1.z, r1.w, N);
      acc += (double)w[j0] + (double)w[j1] + u01_53(r0.x, r0.y) + u01_53(r1.x, r1.y);
    }
    out[i] = acc;
  }
}

template <class T> static float run(uint32_t N, uint32_t B)
{
  T *w; double *out;
  if (hipMalloc(&w, (size_t)N * sizeof(T)) != hipSuccess || hipMalloc(&out, (size_t)N * 8) != hipSuccess) return -1;
  (void)hipMemset(w, 0x11, (size_t)N * sizeof(T));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(chains<T>, dim3(2048), dim3(256), 0, 0, w, N, B, out);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(chains<T>, dim3(2048), dim3(256), 0, 0, w, N, B, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(w); (void)hipFree(out);
  return ms / 50 * 1e3f;
}

int main()
{
  for (uint32_t N : {100000u, 500000u, 1000000u, 4000000u})
    printf("N = %7u, B = 10: gathers from doubles %.1f us | 32-bit words %.1f us | 16-bit words %.1f us\n", N,
           run<double>(N, 10), run<uint32_t>(N, 10), run<uint16_t>(N, 10));
  return 0;
}
