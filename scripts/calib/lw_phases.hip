// Where a group's time goes in logpdf_mfma_wide_kernel, per compute wave (5e5 x 256): the kernel is compiled here with
// -DCUSMC_LW_PHASES; random operands (timing only).
//   hipcc <library flags> -DCUSMC_LW_PHASES scripts/calib/lw_phases.hip -o scripts/calib/lw_phases
#include "../../cusmc_amd/csrc/kernels/logpdf_mfma_wide.hip"

#include <cstdio>
#include <random>
#include <vector>

using namespace cusmc;

int main(int argc, char **argv)
{
  const int d = argc > 1 ? atoi(argv[1]) : 256;
  const int nb = (d + 15) / 16;
  const long N = 128000000L / d;
  std::mt19937_64 gen(1);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> hX((size_t)N * d), M((size_t)d * d, 0.0), hF(mfma_wide_frag_doubles(nb)), z(256, 0.0);
  for (auto &v : hX) v = nd(gen);
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) M[(size_t)i * d + j] = (i == j ? 1.0 : 0.05 * nd(gen));
  mfma_wide_pack_frags(M.data(), d, hF.data());
  double *X, *out, *F, *zz;
  (void)hipMalloc(&X, hX.size() * 8); (void)hipMalloc(&out, N * 8); (void)hipMalloc(&F, hF.size() * 8); (void)hipMalloc(&zz, 2048);
  (void)hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(F, hF.data(), hF.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(zz, z.data(), 2048, hipMemcpyHostToDevice);
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  Epilogue ep{-10.0, 0, 0, 0, 0};
  auto launch = [&]() { return launch_logpdf_mfma_wide(X, N, d, d, true, false, F, zz, zz, ep, out, cus, 0); };
  for (int i = 0; i < 5; ++i) if (launch() != hipSuccess) { printf("launch failed\n"); return 1; }
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> zero(1024 * 8 * 4, 0), h(1024 * 8 * 4);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lw_phases), zero.data(), zero.size() * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) (void)launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_lw_phases), h.size() * 8);
  printf("d = %d (NB = %d), %ld particles: %.1f us per launch; per compute wave, share of its time (mean over %d workgroups):\n", d, nb, N, ms / reps * 1e3, cus);
  printf("  wave  blocks(lo,hi)   k loop   reduction   barrier   after barrier\n");
  for (int w = 0; w < 8; ++w) {
    double s[4] = {0, 0, 0, 0}, tot = 0;
    for (int b = 0; b < cus; ++b) for (int k = 0; k < 4; ++k) { s[k] += (double)h[((size_t)b * 8 + w) * 4 + k]; tot += (double)h[((size_t)b * 8 + w) * 4 + k]; }
    printf("  %d     (%2d,%2d)        %5.1f %%   %5.1f %%     %5.1f %%   %5.1f %%\n", w, wide_lo(nb, w), wide_hi(nb, w), 100 * s[0] / tot, 100 * s[1] / tot, 100 * s[2] / tot, 100 * s[3] / tot);
  }
  return 0;
}
