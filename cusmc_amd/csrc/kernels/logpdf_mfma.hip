// Launchers and host-side fragment packing for kernels/logpdf_mfma_kernel.h (the kernel itself
// and the design notes live there).
#include "logpdf_mfma_kernel.h"

namespace cusmc {

int mfma_num_frags(int nb, bool tri) { return tri ? 4 * nb * (nb + 1) / 2 : 4 * nb * nb; }

// d = 16, 32, ..., 176 with 16-byte aligned rows take the plain kernel; every other d in (16, 176]
// and every other alignment the padded variant (PAD).  d < 16 stays with the generic kernel: a
// single, mostly empty k-block would cost more loads than it saves.  A tile's per-lane byte offset
// is kept in 32 bits.
bool mfma_supported(int d, const void *X, int64_t ldx)
{
  return d >= 16 && d <= kTileKernelMaxDim && ldx < (1L << 24);
}
static bool mfma_needs_pad(int d, const void *X, int64_t ldx)
{
  return d % 16 != 0 || (uintptr_t)X % 16 != 0 || ldx % 2 != 0;
}

// Fragment f (kernel loop order: kb, then s, then cb) holds, for lane l = (j, h):
//   M[16*cb + j][16*kb + pi(s,h)]
void mfma_pack_frags(const double *M, int d, bool tri, double *frags)
{
  const int nb = d / 16;
  int f = 0;
  for (int kb = 0; kb < nb; ++kb)
    for (int s = 0; s < 4; ++s)
      for (int cb = tri ? kb : 0; cb < nb; ++cb, ++f)
        for (int l = 0; l < 64; ++l) {
          const int j = l & 15, h = l >> 4;
          frags[(size_t)f * 64 + l] = M[(size_t)(16 * cb + j) * d + 16 * kb + pi_k(s, h)];
        }
}

template <int NB, bool CENTRED, bool SHIFT, int EPI, bool PAD>
static hipError_t launch_nb(const double *X, int64_t N, int64_t ldx, int d, const double *frags,
                            const double *shift, const double *bias, const Epilogue &ep,
                            double *out, int num_cus, hipStream_t stream)
{
  constexpr int NFRAG = 4 * NB * (NB + 1) / 2;  // lower triangular in both forms
  constexpr int THREADS = mfma_threads<NB>();
  const size_t lds_bytes = (size_t)(32 * NB + 4 + NFRAG * 64) * sizeof(double);  // (the factor passes through LDS in every variant)
  const long num_tiles = (N + 15) / 16;
  auto kern = logpdf_mfma_kernel<NB, CENTRED, SHIFT, 0, EPI, PAD>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  // one persistent workgroup per CU (its waves share the round counter), fewer when there is
  // less work than that
  long blocks = num_cus;
  if (blocks > (num_tiles + 15) / 16) blocks = (num_tiles + 15) / 16;  // >= 16 tiles each
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), lds_bytes, stream, X, (long)N,
                     (long)ldx, frags, shift, bias, ep, out, num_tiles, d);
  return hipGetLastError();
}

// frags: mfma_pack_frags(., ., tri = true, .) of the lower triangular factor.  centred: z = L (x -
// shift); otherwise z = bias + L x (the QL-rotated affine form, cusmc_abi.hip plan_affine()).
hipError_t launch_logpdf_mfma(const double *X, int64_t N, int64_t ldx, int d, bool centred,
                              bool has_shift, const double *frags, const double *shift,
                              const double *bias, const Epilogue &ep, double *out, int num_cus,
                              hipStream_t stream)
{
  if (N <= 0) return hipSuccess;
  const int epi = ep.out_density ? 0 : ep.kind == CUSMC_MVN ? 1 : 2;
  const bool pad = mfma_needs_pad(d, X, ldx);
#define CUSMC_PADV(nb, t, s, l)                                                                   \
  (pad ? launch_nb<nb, t, s, l, true>(X, N, ldx, d, frags, shift, bias, ep, out, num_cus, stream) \
       : launch_nb<nb, t, s, l, false>(X, N, ldx, d, frags, shift, bias, ep, out, num_cus, stream))
#define CUSMC_EPI(nb, t, s) (epi == 1 ? CUSMC_PADV(nb, t, s, 1) : epi == 2 ? CUSMC_PADV(nb, t, s, 2) : CUSMC_PADV(nb, t, s, 0))
#define CUSMC_CASE(nb)                                                                            \
  case nb:                                                                                        \
    if (!centred) return CUSMC_EPI(nb, false, false);                                               \
    return has_shift ? CUSMC_EPI(nb, true, true) : CUSMC_EPI(nb, true, false);
  switch ((d + 15) / 16) {
    CUSMC_CASE(1)
    CUSMC_CASE(2)
    CUSMC_CASE(3)
    CUSMC_CASE(4)
    CUSMC_CASE(5)
    CUSMC_CASE(6)
    CUSMC_CASE(7)
    CUSMC_CASE(8)
#if CUSMC_TILE_MAX_NB >= 9
    CUSMC_CASE(9)
#endif
#if CUSMC_TILE_MAX_NB >= 10
    CUSMC_CASE(10)
#endif
#if CUSMC_TILE_MAX_NB >= 11
    CUSMC_CASE(11)
#endif
  }
#undef CUSMC_CASE
#undef CUSMC_EPI
#undef CUSMC_PADV
  return hipErrorInvalidValue;
}

}  // namespace cusmc
