// Issue cost of the VALU instructions the draw paths are made of (gfx950): cycles per wave instruction on one SIMD,
// measured with 4 waves per SIMD x 8 independent chains per wave (latency hidden), s_memtime around the loop.
//   hipcc -O3 --offload-arch=gfx950 scripts/calib/valu_rate.hip -o scripts/calib/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 512
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long *cycles, unsigned *sink, unsigned seed)
{
  unsigned a[8];
  double f[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; f[i] = 1.0 + 1e-3 * (double)(threadIdx.x + i + seed); }
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) { unsigned long long p; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a[i]), "v"(0xD2511F53u) : "vcc"); a[i] = (unsigned)(p >> 32) ^ (unsigned)p; }
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u));
      if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0x9E3779B9u));
      if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(0.999));
      if (OP == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[i]));
      if (OP == 6) asm volatile("v_rsq_f64 %0, %0" : "+v"(f[i]));
      if (OP == 7) asm volatile("v_sqrt_f64 %0, %0" : "+v"(f[i]));
      if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(0.999));
      if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(0.999));
      if (OP == 10) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      if (OP == 11) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(f[i]) : "v"(1));
      if (OP == 12) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(f[i]));
      if (OP == 13) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x511F53u));
      if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(0x511F53u) : );
      if (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0x511F53u));
      if (OP == 16) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(*(unsigned long long *)&f[i]));
      if (OP == 17) asm volatile("v_mov_b64 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i] ^ (unsigned)__double_as_longlong(f[i]);
  if (s == 0x12345678u) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char *name, unsigned long long *dc, unsigned *ds)
{
  // one block of 256 threads per SIMD would need placement control; instead fill the chip: 256 CUs x 4 blocks of
  // 4 waves = 4 waves per SIMD, every SIMD equally loaded
  const int blocks = 256 * 4;
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, dc, ds, 1u);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, dc, ds, 2u);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= h.size();
  // a SIMD executed 4 waves x REP x 8 instructions in `mean` counter ticks (s_memtime: 100 MHz on gfx950?  report raw)
  printf("%-18s %10.0f ticks per wave for %d instr x 4 waves/SIMD -> %.3f ticks per wave-instruction\n", name, mean, REP * 8,
         mean / (REP * 8 * 4));
}

int main()
{
  unsigned long long *dc; unsigned *ds;
  (void)hipMalloc(&dc, 256 * 4 * 4 * 8); (void)hipMalloc(&ds, 64);
  run<3>("v_xor_b32", dc, ds);
  run<15>("v_add_u32", dc, ds);
  run<14>("v_cndmask_b32", dc, ds);
  run<13>("v_mul_u32_u24", dc, ds);
  run<0>("v_mad_u64_u32(+2)", dc, ds);
  run<1>("v_mul_lo_u32", dc, ds);
  run<2>("v_mul_hi_u32", dc, ds);
  run<16>("v_lshl_add_u64", dc, ds);
  run<17>("v_mov_b64", dc, ds);
  run<4>("v_fma_f64", dc, ds);
  run<8>("v_mul_f64", dc, ds);
  run<9>("v_add_f64", dc, ds);
  run<5>("v_rcp_f64", dc, ds);
  run<6>("v_rsq_f64", dc, ds);
  run<7>("v_sqrt_f64", dc, ds);
  run<10>("v_cvt_f64_u32", dc, ds);
  run<11>("v_ldexp_f64", dc, ds);
  run<12>("v_frexp_mant_f64", dc, ds);
  return 0;
}
