// Experiment: the headline log-pdf kernel (NB = 4, d = 64, centred, no shift, MVN epilogue) with the particle
// rows moved HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`) into a per-wave ring, instead of HBM -> VGPR.
// A/B against the product kernel, interleaved on one box.  Not product code, not a test.
//   hipcc <library flags> scripts/calib/dma_exp.hip cusmc_amd/csrc/build/kernels/logpdf_mfma.o -o scripts/calib/dma_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>

#include "../../cusmc_amd/csrc/launch.h"
#include "../../cusmc_amd/csrc/kernels/logpdf_mfma_kernel.h"

using namespace cusmc;

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ void dma_slab(unsigned lds_addr, unsigned voff, v4u rs, unsigned soff)
{
  if (AUX == 0)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EARLY: the first two tiles are taken by wave number and their DMA is issued before the factor is staged.
template <int NB, int AUX, bool EARLY>
__global__ __launch_bounds__(512) void logpdf_dma_kernel(const double *__restrict__ X, long N, long ldx, const double *__restrict__ frags,
                                                         Epilogue ep, double *__restrict__ out, long num_tiles)
{
  constexpr int THREADS = 512, WAVES = 8;
  constexpr int NFRAG = 4 * NB * (NB + 1) / 2;
  constexpr int SLOT = NB * 2 * 1024;  // bytes per staged tile: slab (kb, h2) = 1 KiB, lane l's 16 bytes at 16 l
  extern __shared__ double lds[];
  double *sF = lds;                                              // NFRAG x 64
  unsigned *sNext = reinterpret_cast<unsigned *>(sF + NFRAG * 64);  // 16 bytes
  char *sRing = reinterpret_cast<char *>(sF + NFRAG * 64 + 2);     // WAVES x 2 x SLOT

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int p = lane & 15, h = lane >> 4;
  const unsigned G = gridDim.x;
  const unsigned nt = (unsigned)num_tiles;
  const unsigned rounds = (nt + G - 1) / G;
  const long last = num_tiles - 1;
  const long tail_rows = N - last * 16;
  const long tile_bytes = 128 * ldx;
  const unsigned voff = (unsigned)((long)p * ldx + 2 * h) * 8u;
  const unsigned full_records = (unsigned)((15 * ldx + 16 * NB) * 8);
  const unsigned last_records = (unsigned)(((tail_rows - 1) * ldx + 16 * NB) * 8);

  const unsigned ring0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)(sRing + (size_t)wave * 2 * SLOT);
  const v2d *rd0 = reinterpret_cast<const v2d *>(sRing + (size_t)wave * 2 * SLOT) + lane;
  const v2d *rd1 = reinterpret_cast<const v2d *>(sRing + (size_t)wave * 2 * SLOT + SLOT) + lane;

  auto issue = [&](unsigned tt, int slot) {
    const long t = tt < nt ? tt : blockIdx.x;  // past the end: re-read the first tile (L2 hit, unused) -- uniform counts
    const unsigned long base = reinterpret_cast<unsigned long>(X) + (unsigned long)(t * tile_bytes);
    v4u rs;
    rs[0] = (unsigned)base;
    rs[1] = (unsigned)(base >> 32) & 0xffffu;
    rs[2] = t == last ? last_records : full_records;
    rs[3] = 0x00020000u;
    const unsigned dst = ring0 + slot * SLOT;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) dma_slab<AUX>(dst + (kb * 2 + h2) * 1024, voff, rs, 128 * kb + 64 * h2);
  };

  unsigned k0 = nt, k1 = nt;
  if (EARLY) {
    const unsigned a = blockIdx.x + (unsigned)wave * G, b = blockIdx.x + (unsigned)(wave + WAVES) * G;
    k0 = ((unsigned)wave < rounds && a < nt) ? a : nt;
    k1 = ((unsigned)(wave + WAVES) < rounds && b < nt) ? b : nt;
    issue(k0, 0);
    issue(k1, 1);
  }

  {
    constexpr int NV = NFRAG * 32;
    constexpr int SC = 8;
    const v2d *g = reinterpret_cast<const v2d *>(frags);
    v2d *l = reinterpret_cast<v2d *>(sF);
    for (int base = 0; base < NV; base += SC * THREADS) {
      v2d tmp[SC];
#pragma unroll
      for (int c = 0; c < SC; ++c) {
        const int i = base + c * THREADS + (int)threadIdx.x;
        if (i < NV) tmp[c] = g[i];
      }
#pragma unroll
      for (int c = 0; c < SC; ++c) {
        const int i = base + c * THREADS + (int)threadIdx.x;
        if (i < NV) l[i] = tmp[c];
      }
    }
  }
  if (threadIdx.x == 0) *sNext = EARLY ? 2 * WAVES : 0;
  __syncthreads();
  double wreg[NFRAG];
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) wreg[f] = sF[f * 64 + lane];

  auto grab = [&]() -> unsigned {
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(sNext, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    const unsigned t = blockIdx.x + k * G;
    return k < rounds && t < nt ? t : nt;
  };

  auto read_slot = [&](const v2d *rd, v2d(&a)[NB][2]) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) a[kb][h2] = rd[(kb * 2 + h2) * 64];
  };

  auto compute_tile = [&](unsigned tu, const v2d(&a_in)[NB][2]) {
    const long t = tu;
    v4d acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = v4d{0.0, 0.0, 0.0, 0.0};
    int f = 0;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = a_in[kb][s >> 1][s & 1];
#pragma unroll
        for (int cb = kb; cb < NB; ++cb, ++f) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(wreg[f], a, acc[cb], 0, 0, 0);
      }
    }
    double q = 0.0;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) q = fma(acc[cb][r], acc[cb][r], q);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (lane < (t == last ? (int)tail_rows : 16)) out[t * 16 + lane] = ep.lognorm - 0.5 * q;
  };

  if (!EARLY) {
    k0 = grab();
    k1 = k0 < nt ? grab() : nt;
    if (k0 < nt) {
      issue(k0, 0);
      issue(k1, 1);
    }
  }
  if (k0 < nt) {
    v2d a[NB][2];
    // VMEM operations of this wave in program order: I(k0) I(k1) | I(k2) S(k0) | I(k3) S(k1) | ...  (I = 2 NB DMA, S = one
    // store); before reading the slot of tile k_n everything younger than I(k_n) may stay in flight.
    auto phase = [&](auto wtag, unsigned kc, const v2d *rd, int slot) -> unsigned {
      constexpr int W = decltype(wtag)::value;
      wait_vm<W>();
      read_slot(rd, a);
      const unsigned kn = grab();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is in registers: it may be overwritten
      issue(kn, slot);
      compute_tile(kc, a);
      return kn;
    };
    unsigned k2 = phase(std::integral_constant<int, 2 * NB>{}, k0, rd0, 0);
    if (k1 < nt) {
      unsigned k3 = phase(std::integral_constant<int, 2 * NB + 1>{}, k1, rd1, 1);
      while (k2 < nt) {
        k0 = phase(std::integral_constant<int, 2 * NB + 2>{}, k2, rd0, 0);
        if (k3 >= nt) break;
        k1 = phase(std::integral_constant<int, 2 * NB + 2>{}, k3, rd1, 1);
        k2 = k0;
        k3 = k1;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land after the wave has gone
}

static constexpr int KT = mfma_threads<4>();
static constexpr size_t KLDS = (size_t)(32 * 4 + 4 + 40 * 64) * 8;
static constexpr size_t DLDS = (size_t)(40 * 64 + 2) * 8 + 8 * 2 * 8192;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
static float timed(F launch, int reps)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}

int main(int argc, char **argv)
{
  const long N = argc > 1 ? atol(argv[1]) : 1000000; const int d = 64;
  std::vector<double> hX((size_t)N * d), M((size_t)d * d, 0.0), frags((size_t)40 * 64), z(64, 0.0);
  unsigned s = 12345;
  for (auto &v : hX) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0 / (1 << 22)); }
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) { s = s * 1664525u + 1013904223u; M[i * d + j] = (i == j) + 0.1 * ((int)(s >> 8) - (1 << 23)) * (1.0 / (1 << 23)); }
  mfma_pack_frags(M.data(), d, true, frags.data());
  double *X, *F, *sh, *bi, *out, *out2;
  CK(hipMalloc(&X, hX.size() * 8)); CK(hipMalloc(&F, frags.size() * 8)); CK(hipMalloc(&sh, 512)); CK(hipMalloc(&bi, 512));
  CK(hipMalloc(&out, N * 8 + 64)); CK(hipMalloc(&out2, N * 8 + 64));
  CK(hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(F, frags.data(), frags.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, z.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(bi, z.data(), 512, hipMemcpyHostToDevice));
  Epilogue ep{-10.0, 0, 0, 0, 0};
  const long tiles = (N + 15) / 16;
  const int blocks = 256;
  CK(hipFuncSetAttribute((const void *)logpdf_mfma_kernel<4, true, false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KLDS));
  CK(hipFuncSetAttribute((const void *)logpdf_dma_kernel<4, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DLDS));
  CK(hipFuncSetAttribute((const void *)logpdf_dma_kernel<4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DLDS));
  CK(hipFuncSetAttribute((const void *)logpdf_dma_kernel<4, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DLDS));
  CK(hipFuncSetAttribute((const void *)logpdf_dma_kernel<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DLDS));
  auto prod = [&]() { hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, 0, 1>), dim3(blocks), dim3(KT), KLDS, 0, X, N, 64L, F, sh, bi, ep, out, tiles, 64); };
  auto d00 = [&]() { hipLaunchKernelGGL((logpdf_dma_kernel<4, 0, false>), dim3(blocks), dim3(512), DLDS, 0, X, N, 64L, F, ep, out2, tiles); };
  auto d20 = [&]() { hipLaunchKernelGGL((logpdf_dma_kernel<4, 2, false>), dim3(blocks), dim3(512), DLDS, 0, X, N, 64L, F, ep, out2, tiles); };
  auto d01 = [&]() { hipLaunchKernelGGL((logpdf_dma_kernel<4, 0, true>), dim3(blocks), dim3(512), DLDS, 0, X, N, 64L, F, ep, out2, tiles); };
  auto d21 = [&]() { hipLaunchKernelGGL((logpdf_dma_kernel<4, 2, true>), dim3(blocks), dim3(512), DLDS, 0, X, N, 64L, F, ep, out2, tiles); };

  // correctness first: every variant must reproduce the product kernel bit for bit
  std::vector<double> ref(N), got(N);
  prod(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
  CK(hipMemcpy(ref.data(), out, N * 8, hipMemcpyDeviceToHost));
  int vi = 0;
  auto check = [&](auto launch, const char *name) {
    (void)hipMemset(out2, 0xff, N * 8);
    launch();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); return false; }
    (void)hipMemcpy(got.data(), out2, N * 8, hipMemcpyDeviceToHost);
    long bad = 0; for (long i = 0; i < N; ++i) bad += memcmp(&ref[i], &got[i], 8) != 0;
    printf("%s: %ld of %ld outputs differ from the product kernel\n", name, bad, N);
    ++vi;
    return bad == 0;
  };
  bool ok = check(d00, "dma") & check(d20, "dma nt") & check(d01, "dma early") & check(d21, "dma nt early");
  if (!ok) { printf("MISMATCH: not timing\n"); return 1; }

  timed(prod, 600);
  std::vector<float> r[5];
  for (int round = 0; round < 9; ++round) {
    r[0].push_back(timed(prod, 200));
    r[1].push_back(timed(d00, 200));
    r[2].push_back(timed(d20, 200));
    r[3].push_back(timed(d01, 200));
    r[4].push_back(timed(d21, 200));
  }
  const char *nm[5] = {"product (HBM->VGPR)", "LDS-DMA ring", "LDS-DMA ring, nt", "LDS-DMA ring, early issue", "LDS-DMA ring, nt, early issue"};
  for (int v = 0; v < 5; ++v) {
    std::sort(r[v].begin(), r[v].end());
    printf("%-32s 9 x 200 launches: median %.1f min %.1f max %.1f us\n", nm[v], r[v][4], r[v][0], r[v][8]);
  }
  // the driver's pattern: 20 timed launches after a short warm-up
  for (int round = 0; round < 3; ++round)
    printf("20-launch bursts: product %.1f | dma %.1f | dma nt %.1f | early %.1f | nt early %.1f\n", timed(prod, 20), timed(d00, 20), timed(d20, 20), timed(d01, 20), timed(d21, 20));
  return 0;
}
