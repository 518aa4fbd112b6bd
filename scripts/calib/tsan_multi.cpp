// Driver for scripts/host_sanitizers.sh tsan: the multi-device filter and resampler with three shards on device 0
// (three host threads + the caller), the library's host code compiled with ThreadSanitizer.
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include "cusmc_hip.h"
static int filter(int d) {
  const uint32_t N = 5003, T = 12;
  std::vector<double> Y(T * d), m0(d, 0.0), I(d * d, 0.0), G(d * d, 0.0), V(d * d, 0.0), W(d * d, 0.0);
  for (int i = 0; i < d; ++i) { I[i * d + i] = 1; G[i * d + i] = 0.9; V[i * d + i] = 0.5; W[i * d + i] = 0.1; }
  for (uint32_t t = 0; t < T; ++t) for (int j = 0; j < d; ++j) Y[t * d + j] = std::sin(0.3 * t + j);
  std::vector<double> X((size_t)T * N * d), w((size_t)T * N); std::vector<uint32_t> a((size_t)T * N);
  int devs[3] = {0, 0, 0};
  int rc = cusmc_pf_run_multi_host(devs, 3, Y.data(), N, d, T, m0.data(), I.data(), I.data(), G.data(), V.data(), W.data(), 0.f,
                                   "metropolis", "mvn", 10, 1.0, 7, X.data(), w.data(), a.data());
  printf("pf_run_multi d=%d rc=%d (%s) w[last]=%g\n", d, rc, cusmc_last_error(), w.back());
  if (rc) return rc;
  std::vector<uint32_t> anc(N);
  rc = cusmc_metropolis_multi_host(devs, 3, w.data() + (size_t)(T - 1) * N, N, 50, 9, 1, 0, anc.data());
  printf("metropolis_multi rc=%d a[0]=%u\n", rc, anc[0]);
  return rc;
}
// d = 2: one fused launch per shard and step (peer reads and stores inside the kernel); d = 12: the three-launch step
// with the sharded gather and the weight copies between the shards
int main() { return filter(2) || filter(12); }
