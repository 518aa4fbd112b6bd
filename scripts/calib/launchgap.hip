// What does one kernel launch cost when kernels follow each other in one stream (the shape of
// bench.py's timed loop)?  Times back-to-back launches of kernels that do (almost) nothing, with the
// headline kernel's launch shape, with and without an 8 MB output written per launch.
// Not product code.  Build: hipcc --offload-arch=gfx950 -O2 -o scripts/calib/launchgap scripts/calib/launchgap.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void k_empty(double *out) {}
__global__ __launch_bounds__(512) void k_write(double *out, long n)
{
  for (long i = blockIdx.x * 512L + threadIdx.x; i < n; i += gridDim.x * 512L) out[i] = (double)i;
}
// occupies the whole register file like the headline kernel (one 512-thread workgroup per CU)
__global__ __launch_bounds__(512) void k_fat(double *out, int n)
{
  double v[100];
#pragma unroll
  for (int i = 0; i < 100; ++i) v[i] = out[(threadIdx.x + i) & 1023];
  double s = 0;
  for (int r = 0; r < n; ++r)
#pragma unroll
    for (int i = 0; i < 100; ++i) { v[i] = v[i] * 1.0000001 + s; s += v[i]; }
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <class F> static float per_launch(F f, int reps)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main()
{
  double *out; if (hipMalloc(&out, 16 << 20) != hipSuccess) return 1;
  (void)hipMemset(out, 0, 16 << 20);
  printf("empty kernel, 256 x 512 threads, back to back: %.2f us per launch\n", per_launch([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, out); }, 2000));
  printf("empty kernel, 1 x 64 threads: %.2f us per launch\n", per_launch([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); }, 2000));
  printf("8 MB written per launch (1e6 doubles): %.2f us per launch\n", per_launch([&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(512), 0, 0, out, 1000000L); }, 2000));
  printf("register-heavy kernel (one workgroup per CU), no work: %.2f us per launch\n", per_launch([&] { hipLaunchKernelGGL(k_fat, dim3(256), dim3(512), 0, 0, out, 0); }, 2000));
  return 0;
}
