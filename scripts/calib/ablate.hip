// Attribution of logpdf_mfma_kernel<4,true> time on the headline shape (N = 1e6, d = 64):
// the product kernel vs. variants with the loads / the MFMAs / the epilogue removed, at several
// grid sizes.  Not product code, not a test.  Build (from repo root):
//   hipcc <library flags of csrc/Makefile> -c scripts/calib/ablate.hip -o /tmp/ablate.o
//   hipcc --offload-arch=gfx950 /tmp/ablate.o cusmc_amd/csrc/build/kernels/logpdf_mfma.o -o scripts/calib/ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <time.h>
#include <vector>
#include <algorithm>

#ifndef KHDR  // -DKHDR=... builds the same harness around another revision of the kernel (A/B runs)
#define KHDR "../../cusmc_amd/csrc/kernels/logpdf_mfma_kernel.h"
#endif
#include KHDR

using namespace cusmc;
static constexpr int KT = mfma_threads<4>();  // launch shape of the kernel under test
static constexpr size_t KLDS = (size_t)(32 * 4 + 4 + 40 * 64) * 8;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ABL>
static float run(const double *X, long N, const double *frags, const double *shift, const double *bias,
                 double *out, int blocks, int reps)
{
  Epilogue ep{-10.0, 0, 0, 0, 0};
  const long tiles = (N + 15) / 16;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, ABL, 1>), dim3(blocks), dim3(KT), KLDS, 0, X, N, 64L, frags, shift, bias, ep, out, tiles);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, ABL, 1>), dim3(blocks), dim3(KT), KLDS, 0, X, N, 64L, frags, shift, bias, ep, out, tiles);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main(int argc, char **argv)
{
  const long N = 1000000; const int d = 64;
  std::vector<double> hX((size_t)N * d), M((size_t)d * d, 0.0), frags((size_t)40 * 64), z(64, 0.0);
  unsigned s = 12345;
  for (auto &v : hX) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0 / (1 << 22)); }
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) { s = s * 1664525u + 1013904223u; M[i * d + j] = (i == j) + 0.1 * ((int)(s >> 8) - (1 << 23)) * (1.0 / (1 << 23)); }
  mfma_pack_frags(M.data(), d, true, frags.data());
  double *X, *F, *sh, *bi, *out;
  CK(hipMalloc(&X, hX.size() * 8)); CK(hipMalloc(&F, frags.size() * 8)); CK(hipMalloc(&sh, 512)); CK(hipMalloc(&bi, 512)); CK(hipMalloc(&out, N * 8 + 32 * 4096 * 8));
  CK(hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(F, frags.data(), frags.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, z.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(bi, z.data(), 512, hipMemcpyHostToDevice));
  if (argc > 1 && !strcmp(argv[1], "persist")) {
    // Are slow workgroups the same ones from launch to launch?  Per-workgroup end times (last wave,
    // 100 MHz stamps) of 8 stamped launches inside a sustained run, and their rank correlation.
    Epilogue ep{-10.0, 0, 0, 0, 0};
    const long tiles = (N + 15) / 16;
    const int L = 8;
    std::vector<std::vector<double>> wg(L, std::vector<double>(256));
    for (int l = 0; l < L; ++l) {
      for (int i = 0; i < 100; ++i)
        hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, 4, 1>), dim3(256), dim3(KT), KLDS, 0, X, N, 64L, F, sh, bi, ep, out, tiles);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> ab(3 * 2048);
      CK(hipMemcpy(ab.data(), out + tiles * 16 + 3 * 2048, ab.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long e0 = ~0ULL; for (int w = 0; w < 2048; ++w) e0 = std::min(e0, ab[3 * w]);
      for (int b = 0; b < 256; ++b) { double m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, (ab[3 * (8 * b + w) + 2] - e0) / 100.0); wg[l][b] = m; }
    }
    double mean[256] = {0};
    for (int l = 0; l < L; ++l) for (int b = 0; b < 256; ++b) mean[b] += wg[l][b] / L;
    auto corr = [&](const std::vector<double> &a, const std::vector<double> &b) {
      double ma = 0, mb = 0; for (int i = 0; i < 256; ++i) { ma += a[i] / 256; mb += b[i] / 256; }
      double sab = 0, saa = 0, sbb = 0; for (int i = 0; i < 256; ++i) { sab += (a[i] - ma) * (b[i] - mb); saa += (a[i] - ma) * (a[i] - ma); sbb += (b[i] - mb) * (b[i] - mb); }
      return sab / std::sqrt(saa * sbb);
    };
    double c = 0; int nc = 0;
    for (int i = 0; i < L; ++i) for (int j = i + 1; j < L; ++j) { c += corr(wg[i], wg[j]); ++nc; }
    printf("mean pairwise correlation of per-workgroup end times over %d launches: %.3f\n", L, c / nc);
    for (int l = 0; l < L; ++l) {
      std::vector<double> s = wg[l]; std::sort(s.begin(), s.end());
      int amax = 0; for (int b = 0; b < 256; ++b) if (wg[l][b] > wg[l][amax]) amax = b;
      printf("launch %d: end min %.1f p50 %.1f p90 %.1f max %.1f (slowest workgroup %d)\n", l, s[0], s[128], s[230], s[255], amax);
    }
    std::vector<int> idx(256); for (int i = 0; i < 256; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return mean[a] > mean[b]; });
    printf("slowest on average:"); for (int i = 0; i < 12; ++i) printf(" %d(%.1f)", idx[i], mean[idx[i]]); printf("\nfastest on average:");
    for (int i = 255; i > 243; --i) printf(" %d(%.1f)", idx[i], mean[idx[i]]); printf("\n");
    double byx[8] = {0}; for (int b = 0; b < 256; ++b) byx[b % 8] += mean[b] / 32;
    printf("mean end by workgroup %% 8:"); for (int i = 0; i < 8; ++i) printf(" %.1f", byx[i]); printf("\n");
    double by32[8] = {0}; for (int b = 0; b < 256; ++b) by32[b / 32] += mean[b] / 32;
    printf("mean end by workgroup / 32:"); for (int i = 0; i < 8; ++i) printf(" %.1f", by32[i]); printf("\n");
    return 0;
  }
  if (argc > 1) {  // quick mode: sustained timing of the product variant only (for A/B between builds)
    std::vector<float> a;
    run<0>(X, N, F, sh, bi, out, 256, 600);
    std::vector<float> b;
    for (int round = 0; round < 15; ++round) {
      a.push_back(run<0>(X, N, F, sh, bi, out, 256, 200));
      b.push_back(run<1>(X, N, F, sh, bi, out, 256, 200));
    }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%s (%d threads, factor in %s): 15 x 200 launches: median %.1f min %.1f | no-loads median %.1f us\n", argv[1], KT,
           mfma_factor_in_regs<4>() ? "VGPRs" : "LDS", a[7], a[0], b[7]);
    return 0;
  }
  {  // sustained vs isolated launches of the product variant at 2 blocks/CU
    Epilogue ep{-10.0, 0, 0, 0, 0};
      const long tiles = (N + 15) / 16;
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, logpdf_mfma_kernel<4, true, false, 0, 1>, KT, KLDS);
    printf("occupancy API (512-thread blocks): %d blocks/CU\n", occ);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256}) {
      for (int reps : {1, 20, 200, 1000}) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < reps; ++i)
          hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, 0, 1>), dim3(blocks), dim3(KT), KLDS, 0, X, N, 64L, F, sh, bi, ep, out, tiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %d, %4d back-to-back launches: %.1f us each\n", blocks, reps, ms / reps * 1e3);
      }
      float tot = 0;
      for (int i = 0; i < 20; ++i) {
        (void)hipDeviceSynchronize();
        struct timespec ts = {0, 3000000}; nanosleep(&ts, nullptr);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, 0, 1>), dim3(blocks), dim3(KT), KLDS, 0, X, N, 64L, F, sh, bi, ep, out, tiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); tot += ms;
      }
      printf("blocks %d, isolated launches (3 ms idle between): %.1f us each\n", blocks, tot / 20 * 1e3);
    }
  }
  for (int mode = 0; mode < 2 && KT == 512; ++mode) {  // in-kernel clock: full kernel, then the no-loads variant (8 waves per workgroup assumed)
    Epilogue ep{-10.0, 0, 0, 0, 0};
      const long tiles = (N + 15) / 16;
    for (int i = 0; i < 300; ++i)
      if (mode == 0) hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, 4, 1>), dim3(256), dim3(KT), KLDS, 0, X, N, 64L, F, sh, bi, ep, out, tiles);
      else hipLaunchKernelGGL((logpdf_mfma_kernel<4, true, false, 5, 1>), dim3(256), dim3(KT), KLDS, 0, X, N, 64L, F, sh, bi, ep, out, tiles);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(3 * 2048);
    CK(hipMemcpy(st.data(), out + tiles * 16, st.size() * 8, hipMemcpyDeviceToHost));
    double csum = 0, rsum = 0, rmax = 0, rmin = 1e30;
    for (int w = 0; w < 2048; ++w) { csum += st[3 * w]; rsum += st[3 * w + 1]; rmax = rmax > st[3*w+1] ? rmax : st[3*w+1]; rmin = rmin < st[3*w+1] ? rmin : st[3*w+1]; }
    {
      double byx[8] = {0}, lo = 0, hi = 0; long tlo = 0, thi = 0;
      for (int w = 0; w < 2048; ++w) {
        double r = st[3 * w + 1] / 100.0; int blk = w / 8;
        byx[blk % 8] += r / 256;
        if ((w % 8) < 4) { lo += r / 1024; tlo += st[3 * w + 2]; } else { hi += r / 1024; thi += st[3 * w + 2]; }
      }
      printf("lifetime by block%%8 (XCD group):"); for (int i = 0; i < 8; ++i) printf(" %.1f", byx[i]); printf("\n");
      printf("waves 0-3: mean %.1f us, %.1f tiles each; waves 4-7: mean %.1f us, %.1f tiles each\n", lo, tlo / 1024.0, hi, thi / 1024.0);
      int hist[12] = {0};
      for (int w = 0; w < 2048; ++w) { int b = (int)((st[3 * w + 1] / 100.0 - 70) / 2); if (b < 0) b = 0; if (b > 11) b = 11; ++hist[b]; }
      printf("lifetime histogram 70..94us step 2:"); for (int i = 0; i < 12; ++i) printf(" %d", hist[i]); printf("\n");
    }
    {  // timeline of the LAST launch from the absolute 100 MHz stamps (10 ns units)
      std::vector<unsigned long long> ab(3 * 2048);
      CK(hipMemcpy(ab.data(), out + tiles * 16 + 3 * 2048, ab.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long e0 = ~0ULL; for (int w = 0; w < 2048; ++w) e0 = std::min(e0, ab[3 * w]);
      double entry_max = 0, pro = 0, end_mean = 0, end_max = 0, end_min = 1e30; std::vector<double> ends;
      for (int w = 0; w < 2048; ++w) {
        const double en = (ab[3 * w] - e0) / 100.0, st = (ab[3 * w + 1] - e0) / 100.0, ed = (ab[3 * w + 2] - e0) / 100.0;
        entry_max = std::max(entry_max, en); pro += (st - en) / 2048; end_mean += ed / 2048; end_max = std::max(end_max, ed); end_min = std::min(end_min, ed);
        ends.push_back(ed);
      }
      std::sort(ends.begin(), ends.end());
      printf("timeline (us from first wave entry): last entry %.2f, prologue mean %.2f, wave end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f mean %.1f\n",
             entry_max, pro, end_min, ends[204], ends[1024], ends[1843], end_max, end_mean);
      double wgend[256]; for (int b = 0; b < 256; ++b) { wgend[b] = 0; for (int w = 0; w < 8; ++w) wgend[b] = std::max(wgend[b], (ab[3 * (8 * b + w) + 2] - e0) / 100.0); }
      std::sort(wgend, wgend + 256);
      printf("workgroup end (last wave): min %.1f p50 %.1f max %.1f\n", wgend[0], wgend[128], wgend[255]);
    }
    printf("[%s] in-kernel clock (300 launches)", mode == 1 ? "no-loads" : "full"); printf(": %.2f GHz; wave lifetime %.1f us mean, %.1f min, %.1f max\n",
           csum / rsum * 0.1, rsum / 2048 / 100.0, rmin / 100.0, rmax / 100.0);
  }
  printf("blocks/CU |  full  | no-loads | no-mfma | no-epilogue-reduce   (us per launch, N=1e6 d=64)\n");
  for (int per : {1}) {
    const int blocks = 256 * per;
    printf("   %d      | %6.1f | %6.1f   | %6.1f  | %6.1f\n", per, run<0>(X, N, F, sh, bi, out, blocks, 20),
           run<1>(X, N, F, sh, bi, out, blocks, 20), run<2>(X, N, F, sh, bi, out, blocks, 20),
           run<3>(X, N, F, sh, bi, out, blocks, 20));
  }
  return 0;
}
