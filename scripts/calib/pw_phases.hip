// Where a group's time goes in propagate_wide_kernel (5e5 x 256, dense G and Q): the kernel is compiled here with
// -DCUSMC_PW_PHASES, which makes wave 0 of every workgroup add the s_memtime ticks between its phase boundaries
// to g_pw_phases.  Random operands (timing only).
//   hipcc <library flags> -DCUSMC_PW_PHASES scripts/calib/pw_phases.hip -o scripts/calib/pw_phases
#include "../../cusmc_amd/csrc/kernels/propagate_mfma_wide.hip"

#include <cstdio>
#include <random>
#include <vector>

using namespace cusmc;

int main(int argc, char **argv)
{
  const int kind = argc > 1 ? atoi(argv[1]) : CUSMC_MVN;
  const int d = 256, nb = 16;
  const uint32_t N = 500000;
  std::mt19937_64 gen(1);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> hX((size_t)N * d), hF((size_t)nb * 4 * nb * 64);
  for (auto &v : hX) v = nd(gen);
  for (auto &v : hF) v = 0.05 * nd(gen);
  std::vector<uint32_t> ha(N);
  for (auto &v : ha) v = (uint32_t)(gen() % N);
  double *X, *out, *fq, *fg; uint32_t *a;
  (void)hipMalloc(&X, hX.size() * 8); (void)hipMalloc(&out, hX.size() * 8); (void)hipMalloc(&fq, hF.size() * 8); (void)hipMalloc(&fg, hF.size() * 8);
  (void)hipMalloc(&a, N * 4);
  (void)hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(fq, hF.data(), hF.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(fg, hF.data(), hF.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(a, ha.data(), N * 4, hipMemcpyHostToDevice);
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  auto launch = [&](uint32_t step) {
    return launch_propagate_mfma_wide(kind, 4.0f, X, a, fq, fg, 1, d, 1.0, 1234, step, 2u, 0, N, out, cus, 0);
  };
  for (int i = 0; i < 3; ++i) if (launch(i) != hipSuccess) { printf("launch failed\n"); return 1; }
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> z(8 * 1024, 0), h(8 * 1024);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_phases), z.data(), z.size() * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) (void)launch(10 + i);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pw_phases), h.size() * 8);
  const char *name[8] = {"", "gather issue + draws -> slabs", "barrier (slabs complete)", "product Q Xi", "[Student-t scaling] barrier, rows -> slabs",
                         "barrier + product G X", "epilogue (stores)", "barrier (slabs free)"};
  double tot = 0, sum[8] = {0};
  for (int b = 0; b < cus; ++b) for (int k = 1; k < 8; ++k) { sum[k] += (double)h[8 * b + k]; tot += (double)h[8 * b + k]; }
  printf("%s, 5e5 x 256 dense: %.1f us per launch; share of wave 0's time per phase (mean over %d workgroups):\n", kind == CUSMC_MVT ? "Student-t" : "Normal", ms / reps * 1e3, cus);
  for (int k = 1; k < 8; ++k) printf("  %-44s %5.1f %%\n", name[k], 100.0 * sum[k] / tot);
  return 0;
}
