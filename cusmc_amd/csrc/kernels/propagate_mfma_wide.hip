// Proposal draws for 128 < d <= 256 (NB = ceil(d/16) = 9 .. 16 blocks) on the f64 matrix cores:
//     X_out^T = Q Xi^T [.* C] + G X_gathered^T          (propagate_K, src/mcmc.cpp:90-160)
//     X_out^T = Q Xi^T [.* C] + m0                       (initialize / R-level draws, src/mcmc.cpp:44-88)
// Replaces, for these d, the one-workgroup-per-particle propagate_row_kernel of round 1 (every particle
// re-read both d x d factors from L2: 1 MB per particle at d = 256) and, on the reference's side, the
// one-block-per-particle-per-tile GEMV pair mvn_sample_Gmu_kernel / mvn_sample_kernel
// (src/mvn_dist.cu.cpp:33-172; mvt twins src/mvt_dist.cu.cpp:84-223).
//
// The factors (512 KB each at d = 256) fit neither LDS nor registers, so -- as in logpdf_mfma_wide.hip --
// the OUTPUT dimension is split over the eight waves of a workgroup and the B operands of a group of 32
// particles (two 16-particle tiles) are staged once in LDS for all of them:
//   fill      every wave fills 1/8 of the group's operand slabs.  Slab (kb, h2, t) holds, for lane (p, h),
//             the operands of k-steps 2 h2 and 2 h2 + 1 of k-block kb for particle 16 t + p: columns
//             16 kb + 8 h2 + 2 h and + 1 -- exactly ONE Box-Muller pair (Philox block sub = 8 kb + 4 h2 + h:
//             the RNG contract keys a block by component pair), resp. one 16-byte piece of the ancestor's
//             row.  A lane writes its 16 bytes at slab + 16 lane; a compute wave reads a slab back with
//             one conflict-free ds_read_b128 per lane.
//   multiply  wave w owns output blocks w and w + 8 for both tiles; the A fragments of its blocks stream
//             from L2 (mfma_pack_frags order, 512 contiguous bytes per fragment) one k-block ahead of
//             their use: 2 blocks x 2 tiles x 4 NB MFMAs per factor per group and wave.
//   epilogue  lane (p, h), register r holds output dim 16 cb + h + 4 r of particle p: Student-t scaling by
//             sqrt(nu / chi2) per component (src/statistics.cc.cpp:385-386, 411; chi_square_batch, smallops.h),
//             + G x (second accumulator), + g .* x (diagonal G) or + m0, stored 8 bytes per lane.
// Two barriers per group.  The fill (RNG: VALU) and the multiply (MFMA) of one workgroup do not overlap;
// an f64 MFMA blocks VALU issue on its SIMD anyway.  (Tried: the next group's ancestor
// rows fetched during the current multiply, held in 16 VGPRs -- no gain for the Normal kernel, 3163 -> 3244 us
// at 5e5 x 256 on another box, and the Student-t one spills: 4857 -> 5235 us.)  Algorithmic bytes per particle
// 16 d + 4; flops 2 x 2 x (16 NB)^2 (dense G) -- MFMA-bound: 1.7 ms for 5e5 x 256 at the 77.7 TF peak.
//
// d that is not a multiple of 16 runs with the factors zero-padded to 16 NB on the host (PAD): normals
// for pairs past d are not drawn, gathered columns past d are zeroed, outputs past d are not stored.
#include "smallops.h"

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));  // rows are 8-byte aligned in general

constexpr int kWideTiles = 2;          // 16-particle tiles per group
constexpr int kWideGroup = 16 * kWideTiles;

bool propagate_mfma_wide_supported(int d, const void *X_prev, const void *X_out)
{
  return d > 128 && d <= 256 && (uintptr_t)X_prev % 8 == 0 && (uintptr_t)X_out % 8 == 0;
}

size_t propagate_wide_lds_bytes(int nb, bool has_g) { return (size_t)nb * 2 * kWideTiles * 1024 * (has_g ? 2 : 1); }

// MODE 0: x = [diag(c)] Q xi + m0     (tail = m0, d doubles)
//      1: x = [diag(c)] Q xi + G x_prev[a]   (tail = fragments of G)
//      4: x = [diag(c)] Q xi + g .* x_prev[a]   (tail = diag(G), d doubles)
template <int NB, bool MVT, int MODE, bool PAD>
__global__ __launch_bounds__(512) void propagate_wide_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ fragsQ, const double *__restrict__ tail, int d, double scale, uint32_t k0,
    uint32_t k1, uint32_t step, uint32_t domain, uint32_t first, uint32_t count, double *__restrict__ X_out,
    long num_groups)
{
  constexpr int T = kWideTiles;
  constexpr bool HAS_G = MODE == 1;
  constexpr int SLABS = NB * 2 * T;
  extern __shared__ double lds[];
  double *sXi = lds;                 // SLABS x 128 doubles
  double *sXg = lds + SLABS * 128;   // the same for the gathered rows (HAS_G)

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int p = lane & 15, h = lane >> 4;
  const ChiSquare cs = chi_setup(MVT ? nu : 2.0f);
  // this wave's output blocks: w, and w + 8 if there is one
  const int cb0 = w, cb1 = w + 8;
  const bool two = cb1 < NB;

  for (long g = blockIdx.x; g < num_groups; g += gridDim.x) {
    const long base = g * kWideGroup;  // first local row of the group
    // ---- fill: this wave's share of the slabs ------------------------------------------------------
    for (int sl = w; sl < SLABS; sl += 8) {
      const int t = sl % T, h2 = (sl / T) & 1, kb = sl / (2 * T);  // (wave-uniform)
      const long local = base + 16 * t + p;
      const bool live = local < (long)count;
      const uint32_t gi = first + (uint32_t)(live ? local : (long)count - 1);
      const int col = 16 * kb + 8 * h2 + 2 * h;  // the pair's first column
      double z0 = 0.0, z1 = 0.0;
      if (!PAD || col < d) {
        normal_pair(philox4x32_10(gi, (uint32_t)(8 * kb + 4 * h2 + h), step, domain, k0, k1), z0, z1);
        z0 *= scale;
        z1 = (!PAD || col + 1 < d) ? z1 * scale : 0.0;
      }
      reinterpret_cast<v2d *>(sXi + sl * 128)[lane] = v2d{z0, z1};
      if constexpr (HAS_G) {
        const uint32_t anc = a ? a[live ? local : (long)count - 1] : gi;
        const double *row = X_prev + (long)anc * d;
        v2d x;
        if (!PAD || col + 1 < d) {
          x = *reinterpret_cast<const v2d_a8 *>(row + col);
        } else {
          x = v2d{col < d ? row[col] : 0.0, 0.0};
        }
        reinterpret_cast<v2d *>(sXg + sl * 128)[lane] = x;
      }
    }
    __syncthreads();
    // ---- multiply: blocks cb0 (and cb1) x both tiles -------------------------------------------------
    v4d accQ[2][T], accG[2][T];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < T; ++t) accQ[b][t] = accG[b][t] = v4d{0.0, 0.0, 0.0, 0.0};
    auto product = [&](const double *__restrict__ frags, const double *sB, v4d(&acc)[2][T]) {
      // fragment (kb, s, cb) is at ((kb 4 + s) NB + cb) x 64.  Both operands of k-block kb + 1 -- eight fragment
      // values from L2, four 16-byte pieces from LDS -- are requested before the MFMAs of k-block kb are issued;
      // two register sets swap roles (loop unrolled by two: no copies).  -5 % (d = 256) to -7 % (d = 192)
      // against fetching the LDS pieces at their use.
      double wa[4][2], wb[4][2];
      v2d xa[2][T], xb[2][T];
      auto load_w = [&](int kb, double(&dst)[4][2]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          dst[s][0] = frags[(size_t)((kb * 4 + s) * NB + cb0) * 64 + lane];
          dst[s][1] = two ? frags[(size_t)((kb * 4 + s) * NB + cb1) * 64 + lane] : 0.0;
        }
      };
      auto load_x = [&](int kb, v2d(&dst)[2][T]) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int t = 0; t < T; ++t) dst[h2][t] = reinterpret_cast<const v2d *>(sB + ((kb * 2 + h2) * T + t) * 128)[lane];
      };
      auto kblock = [&](int kb, double(&wc)[4][2], v2d(&xc)[2][T], double(&wn)[4][2], v2d(&xn)[2][T]) {
        if (kb + 1 < NB) {
          load_w(kb + 1, wn);
          load_x(kb + 1, xn);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const double bv = xc[s >> 1][t][s & 1];
            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s][0], bv, acc[0][t], 0, 0, 0);
            if (two) acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s][1], bv, acc[1][t], 0, 0, 0);
          }
      };
      load_w(0, wa);
      load_x(0, xa);
      int kb = 0;
#pragma unroll 1
      for (; kb + 1 < NB; kb += 2) {
        kblock(kb, wa, xa, wb, xb);
        kblock(kb + 1, wb, xb, wa, xa);
      }
      if (kb < NB) kblock(kb, wa, xa, wb, xb);
    };
    product(fragsQ, sXi, accQ);
    if constexpr (HAS_G) product(tail, sXg, MVT ? accG : accQ);  // (Normal: one accumulator takes both products)
    // ---- epilogue ------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const long local = base + 16 * t + p;
      if (local >= (long)count) continue;
      const uint32_t gi = first + (uint32_t)local;
      // the lane's eight components of this particle: c = 4 b + r  ->  j = 16 cb_b + h + 4 r
      auto jof = [&](int c) { return 16 * ((c >> 2) ? cb1 : cb0) + h + 4 * (c & 3); };
      auto ok = [&](int c) { return ((c >> 2) == 0 || two) && (!PAD || jof(c) < d); };
      double chi[MVT ? 8 : 1];
      if constexpr (MVT) chi_square_batch<8>(cs, gi, step, k0, k1, jof, ok, chi);
      const double *xrow = nullptr;
      if constexpr (MODE == 4) xrow = X_prev + (long)(a ? a[local] : gi) * d;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b == 1 && !two) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * (b ? cb1 : cb0) + h + 4 * r;
          if (PAD && j >= d) continue;
          double v = accQ[b][t][r];
          if constexpr (MVT) {
            v *= sqrt((double)nu / chi[4 * b + r]);
            if (HAS_G) v += accG[b][t][r];
          }
          if constexpr (MODE == 0) v += tail[j];
          if constexpr (MODE == 4) v += fma(tail[j], xrow[j], 0.0);  // (the one non-zero term of the dense kernels' sum)
          X_out[local * d + j] = v;
        }
      }
    }
    __syncthreads();  // every wave has read the slabs: the next group may overwrite them
  }
}

template <int NB, bool MVT, int MODE, bool PAD>
static hipError_t launch_pw(float nu, const double *X_prev, const uint32_t *a, const double *fragsQ, const double *tail,
                            int d, double scale, uint64_t seed, uint32_t step, uint32_t domain, uint32_t first,
                            uint32_t count, double *X_out, int num_cus, hipStream_t stream)
{
  const size_t lds_bytes = propagate_wide_lds_bytes(NB, MODE == 1);
  auto kern = propagate_wide_kernel<NB, MVT, MODE, PAD>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  const long num_groups = ((long)count + kWideGroup - 1) / kWideGroup;
  long blocks = num_cus;
  if (blocks > num_groups) blocks = num_groups;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, stream, nu, X_prev, a, fragsQ, tail, d, scale,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out, num_groups);
  return hipGetLastError();
}

// fragsQ: dense mfma_pack_frags image of Q zero-padded to 16 NB; tail: the same of G (mode 1), diag(G)
// (mode 4) or m0 (mode 0), both padded to 16 NB entries where they are vectors.
hipError_t launch_propagate_mfma_wide(int kind, float nu, const double *X_prev, const uint32_t *a, const double *fragsQ,
                                      const double *tail, int mode, int d, double scale, uint64_t seed, uint32_t step,
                                      uint32_t domain, uint32_t first, uint32_t count, double *X_out, int num_cus,
                                      hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  const bool mvt = kind == CUSMC_MVT;
  const bool pad = d % 16 != 0;
#define CUSMC_PW_MODE(nb, m)                                                                                              \
  (mvt ? (pad ? launch_pw<nb, true, m, true>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)   \
              : launch_pw<nb, true, m, false>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)) \
       : (pad ? launch_pw<nb, false, m, true>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)  \
              : launch_pw<nb, false, m, false>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)))
#define CUSMC_PW_CASE(nb) \
  case nb:                \
    return mode == 0 ? CUSMC_PW_MODE(nb, 0) : mode == 1 ? CUSMC_PW_MODE(nb, 1) : CUSMC_PW_MODE(nb, 4);
  switch ((d + 15) / 16) {
    CUSMC_PW_CASE(9)
    CUSMC_PW_CASE(10)
    CUSMC_PW_CASE(11)
    CUSMC_PW_CASE(12)
    CUSMC_PW_CASE(13)
    CUSMC_PW_CASE(14)
    CUSMC_PW_CASE(15)
    CUSMC_PW_CASE(16)
  }
#undef CUSMC_PW_CASE
#undef CUSMC_PW_MODE
  return hipErrorInvalidValue;
}

}  // namespace cusmc
