// Hardware calibration for the fp64 log-pdf roofline (not product code, not a test):
//   1. v_mfma_f64_16x16x4_f64 issue rate per SIMD (1..3 waves/SIMD)
//   2. v_fma_f64 VALU rate per SIMD
//   3. both at once from different waves of one SIMD (are the pipes really separate for f64?)
//   4. HBM streaming read rate: flat 16 B/lane, and the MFMA-A-fragment pattern (16 rows x 64 B)
// Build: hipcc --offload-arch=gfx950 -O3 -o calib calib.hip ; run: ./calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode bit0: waves with (wave & 1) == 0 run MFMA; bit1: others run VALU.  split=0: all waves same.
// random operands + in-kernel clock: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime)
__global__ __launch_bounds__(1024) void rate_rand_kernel(const double *in, double *out, int iters, unsigned long long *clk)
{
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096]; }
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c3, 0, 0, 0);
    if ((i & 255) == 255) { c0 *= 1e-3; c1 *= 1e-3; c2 *= 1e-3; c3 *= 1e-3; }  // keep finite
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// The kernel's MFMA stream in isolation: 40 MFMAs per "tile" in the triangular pattern, 40
// distinct factor registers, 16 distinct particle operands, 4 accumulators; EPI adds the 16-FMA
// square-sum.  Operands live in registers: no memory, no LDS.
template <int EPI>
__global__ __launch_bounds__(512) void stream_kernel(const double *in, double *out, int tiles)
{
  double w[40], a[16];
  for (int i = 0; i < 40; ++i) w[i] = in[(threadIdx.x + 64 * i) % 4096];
  for (int i = 0; i < 16; ++i) a[i] = in[(threadIdx.x * 3 + 64 * i) % 4096];
  double tot = 0;
  for (int t = 0; t < tiles; ++t) {
    v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int f = 0;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = kb; cb < 4; ++cb, ++f)
          acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[f], a[kb * 4 + s], acc[cb], 0, 0, 0);
    if (EPI) {
      double q = 0;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) q = fma(acc[cb][r], acc[cb][r], q);
      tot += q;
    } else {
      tot += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
    asm volatile("" : "+v"(a[0]));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void rate_kernel(double *out, int iters, int mode)
{
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode >= 2 && (wave & 4) == 0);
  const bool do_valu = (mode == 1) || (mode == 2 && (wave & 4) != 0);
  const bool do_int = (mode == 3 && (wave & 4) != 0) || mode == 5;
  const bool do_lds = (mode == 4 && (wave & 4) != 0) || mode == 6;
  const bool do_lds_pure = (mode == 7 && (wave & 4) != 0) || mode == 8;
  const bool do_gld_pure = (mode == 9 && (wave & 4) != 0) || mode == 10;
  __shared__ double sl[1024];
  sl[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (do_int) {  // 64 32-bit VALU ops per iteration
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x0 = x0 * 3 + x1; x1 = x1 * 3 + x2; x2 = x2 * 3 + x3; x3 = x3 * 3 + x4;
        x4 = x4 * 3 + x5; x5 = x5 * 3 + x6; x6 = x6 * 3 + x7; x7 = x7 * 3 + x0;
      }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    return;
  }
  if (do_lds_pure) {  // 16 ds_read_b64 per iteration, NO VALU: results only kept alive
    unsigned addr = (threadIdx.x & 63) * 8;
    for (int i = 0; i < iters; ++i) {
      double r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   "ds_read_b64 %0, %8 offset:4096\n ds_read_b64 %1, %8 offset:4608\n ds_read_b64 %2, %8 offset:5120\n ds_read_b64 %3, %8 offset:5632\n"
                   "ds_read_b64 %4, %8 offset:6144\n ds_read_b64 %5, %8 offset:6656\n ds_read_b64 %6, %8 offset:7168\n ds_read_b64 %7, %8 offset:7680\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr) : "memory");
    }
    return;
  }
  if (do_gld_pure) {  // 8 global_load_dwordx2 (L2/L1 hits) per iteration, no VALU
    const double *gp = out + (threadIdx.x & 63);
    for (int i = 0; i < iters / 2; ++i) {
      double r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("global_load_dwordx2 %0, %8, off\n global_load_dwordx2 %1, %8, off offset:512\n global_load_dwordx2 %2, %8, off offset:1024\n global_load_dwordx2 %3, %8, off offset:1536\n"
                   "global_load_dwordx2 %4, %8, off offset:2048\n global_load_dwordx2 %5, %8, off offset:2560\n global_load_dwordx2 %6, %8, off offset:3072\n global_load_dwordx2 %7, %8, off offset:3584\n"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(gp) : "memory");
    }
    return;
  }
  if (do_lds) {  // 16 ds_read_b64 per iteration
    double acc = 0; int idx = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { acc += sl[(idx + u * 64) & 1023]; }
      asm volatile("" : "+v"(idx));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    return;
  }
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  if (do_mfma) {
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (do_valu) {
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // 64 FMAs per iteration
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
      }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}

__global__ __launch_bounds__(256) void read_flat(const v2d *__restrict__ x, long n16, double *out)
{
  double s = 0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    v2d v = x[i];
    s += v[0] + v[1];
  }
  if (s == 1.2345) out[0] = s;
}

// the A-fragment pattern of logpdf_mfma_kernel<4>: per wave, per tile of 16 rows x 512 B,
// 8 loads of 16 B per lane: lane (p = l&15, h = l>>4) reads row p, bytes 128*kb + 16*h (+64).
__global__ __launch_bounds__(256) void read_frag(const double *__restrict__ X, long tiles, double *out)
{
  const int lane = threadIdx.x & 63, p = lane & 15, h = lane >> 4;
  const long nw = (long)gridDim.x * 4;
  double s = 0;
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += nw) {
    const double *src = X + (t * 16 + p) * 64 + 2 * h;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      v2d u = *(const v2d *)(src + 16 * kb), v = *(const v2d *)(src + 16 * kb + 8);
      s += u[0] + u[1] + v[0] + v[1];
    }
  }
  if (s == 1.2345) out[0] = s;
}

static float time_launch(void (*launch)(void), int reps)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

static double *g_out; static int g_iters, g_mode, g_blocks, g_threads;
static void launch_rate() { hipLaunchKernelGGL(rate_kernel, dim3(g_blocks), dim3(g_threads), 0, 0, g_out, g_iters, g_mode); }
static const double *g_x; static long g_n; static int g_rb;
static void launch_flat() { hipLaunchKernelGGL(read_flat, dim3(g_rb), dim3(256), 0, 0, (const v2d *)g_x, g_n / 2, g_out); }
static void launch_frag() { hipLaunchKernelGGL(read_frag, dim3(g_rb), dim3(256), 0, 0, g_x, g_n / 1024, g_out); }

int main()
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  CK(hipMalloc(&g_out, sizeof(double) * 1024 * 1024 * 4));
  g_iters = 20000;
  for (int wps = 1; wps <= 4; ++wps) {  // waves per SIMD
    g_blocks = cus; g_threads = 256 * wps;
    g_mode = 0;
    float ms = time_launch(launch_rate, 3);
    double mf = (double)g_blocks * (g_threads / 64) * g_iters * 4;  // MFMAs
    printf("mfma f64: %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", wps, ms,
           mf * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)g_iters * 4 * wps));
    g_mode = 1;
    ms = time_launch(launch_rate, 3);
    double fm = (double)g_blocks * g_threads * g_iters * 64;  // lane-FMAs
    printf("valu f64: %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.2f cycles/wave-FMA/SIMD @2.4GHz\n", wps, ms,
           fm * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)g_iters * 64 * wps));
  }
  // mixed: 8 waves per block = 2 per SIMD; waves 0-3 MFMA, 4-7 VALU
  g_blocks = cus; g_threads = 512; g_mode = 2;
  {
    float ms = time_launch(launch_rate, 3);
    double flop = (double)g_blocks * 4 * g_iters * 4 * 2048 + (double)g_blocks * 256 * g_iters * 64 * 2;
    printf("mixed (1 MFMA wave + 1 VALU wave per SIMD): %.3f ms, %.1f TFLOP/s total\n", ms, flop / ms / 1e9);
  }
  for (int m = 3; m <= 10; ++m) {
    g_blocks = cus; g_threads = (m <= 4 || m == 7 || m == 9) ? 512 : 256; g_mode = m;
    float ms = time_launch(launch_rate, 3);
    const char *nm[] = {"", "", "", "mixed MFMA wave + int-VALU wave per SIMD", "mixed MFMA wave + LDS-read wave per SIMD", "int-VALU alone (1 wave/SIMD)", "LDS-read alone (1 wave/SIMD)",
                        "mixed MFMA wave + PURE ds_read wave (no VALU)", "pure ds_read alone", "mixed MFMA wave + PURE global_load wave (no VALU)", "pure global_load alone"};
    printf("%s: %.3f ms\n", nm[m], ms);
  }
  {  // random operands, clock
    std::vector<double> h(4096); unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((int)(sd >> 8) - (1 << 23)) * (1.0 / (1 << 22)); }
    double *in; unsigned long long *clk; CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&clk, 16 * cus));
    CK(hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    for (int wps = 1; wps <= 2; ++wps) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_rand_kernel, dim3(cus), dim3(256 * wps), 0, 0, in, g_out, 100000, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(2 * cus); CK(hipMemcpy(hc.data(), clk, 16 * cus, hipMemcpyDeviceToHost));
        double ghz = (double)hc[0] / (double)hc[1] * 0.1;
        printf("mfma f64 RANDOM operands, %d waves/SIMD: %.3f ms, %.1f TFLOP/s, in-kernel clock %.2f GHz\n", wps, ms,
               (double)cus * 4 * wps * 100000 * 4 * 2048 / ms / 1e9, ghz);
      }
    }
  }
  {
    std::vector<double> h(4096); unsigned sd = 7;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((int)(sd >> 8) - (1 << 23)) * (1.0 / (1 << 26)); }
    double *in; CK(hipMalloc(&in, 4096 * 8)); CK(hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = 2000;
    for (int threads : {256, 512}) for (int epi = 0; epi < 2; ++epi) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (epi) hipLaunchKernelGGL(stream_kernel<1>, dim3(cus), dim3(threads), 0, 0, in, g_out, tiles);
        else hipLaunchKernelGGL(stream_kernel<0>, dim3(cus), dim3(threads), 0, 0, in, g_out, tiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("kernel-like MFMA stream, %d waves/SIMD, epilogue %d: %.1f cycles per tile per SIMD @2.4GHz (2592 = MFMA only)\n",
                        threads / 256, epi, ms * 1e-3 * 2.4e9 / (tiles * (threads / 256)));
      }
    }
  }
  g_n = 64L * 1000 * 1000;  // doubles: 512 MB
  double *x; CK(hipMalloc(&x, g_n * 8)); CK(hipMemset(x, 0, g_n * 8)); g_x = x;
  for (int per : {2, 4, 8, 12, 16, 32}) {
    g_rb = cus * per;
    float a = time_launch(launch_flat, 10), b = time_launch(launch_frag, 10);
    printf("read 512 MB, %2d blocks/CU: flat %.1f us = %.2f TB/s | fragment-pattern %.1f us = %.2f TB/s\n", per,
           a * 1e3, g_n * 8 / a / 1e9, b * 1e3, g_n * 8 / b / 1e9);
  }
  return 0;
}
