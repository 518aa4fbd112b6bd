// Stall attribution for logpdf_mfma_wide_kernel<16,true,false> (d = 256, N = 5e5).  Not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
namespace cusmc {
size_t mfma_wide_frag_doubles(int nb);
void mfma_wide_pack_frags(const double *M, int d, double *frags);
int wide_occupancy_probe();
hipError_t launch_wide_ablate(int abl, const double *X, int64_t N, const double *frags, const double *zeros, double *out, int blocks, hipStream_t stream);
}
int main() {
  const long N = 500000; const int d = 256;
  std::vector<double> M((size_t)d * d, 0.0), hX((size_t)N * d);
  unsigned s = 1; for (auto &v : hX) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0 / (1 << 22)); }
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) { s = s * 1664525u + 1013904223u; M[i * d + j] = (i == j) + 0.05 * ((int)(s >> 8) - (1 << 23)) * (1.0 / (1 << 23)); }
  std::vector<double> fr(cusmc::mfma_wide_frag_doubles(16)); cusmc::mfma_wide_pack_frags(M.data(), d, fr.data());
  double *X, *F, *Z, *out; hipMalloc(&X, hX.size() * 8); hipMalloc(&F, fr.size() * 8); hipMalloc(&Z, 4096); hipMalloc(&out, N * 8);
  hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice); hipMemcpy(F, fr.data(), fr.size() * 8, hipMemcpyHostToDevice); hipMemset(Z, 0, 4096);
  printf("occupancy API: %d workgroups/CU\n", cusmc::wide_occupancy_probe());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *nm[] = {"full", "no W refetch", "no X refetch", "neither"};
  for (int blocks : {256}) for (int a = 0; a < 4; ++a) {
    for (int i = 0; i < 3; ++i) cusmc::launch_wide_ablate(a, X, N, F, Z, out, blocks, 0);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) cusmc::launch_wide_ablate(a, X, N, F, Z, out, blocks, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %d %-14s %.1f us  (MFMA floor 448 us @2.4GHz)\n", blocks, nm[a], ms / 20 * 1e3);
  }
  return 0;
}
