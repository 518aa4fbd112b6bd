// Proposal draws for 16 <= d <= 128 on the f64 matrix cores: gather + RNG + two dense mat-vecs
// per particle become   X_out^T = Q Xi^T [.* C] + G X_gathered^T   (d x d times d x 16 per tile).
// Same contract, RNG counters and reference functions as kernels/propagate.hip
// (propagate_K: src/mcmc.cpp:112-140 -> sample(): src/statistics.cc.cpp:224-259, 355-412; replaces
// mvn_sample_*_kernel, src/mvn_dist.cu.cpp:15-172, and the mvt twins src/mvt_dist.cu.cpp:63-223).
//
// Mapping (as kernels/logpdf_mfma_kernel.h): one wave = 16 particles per tile.
//   A operand  rows of Q, then rows of G: lane (j, h) holds M[16 cb + j][16 kb + pi(s,h)], packed on
//              the host (mfma_pack_frags, dense) and staged once per workgroup in LDS
//              (2 x 4 NB^2 fragments: 64 KB at d = 64).
//   B operand  lane (p, h) holds, for particle p, the k-values 16 kb + pi(s,h):
//                xi: generated IN that layout -- k-pairs (2h, 2h+1) and (8+2h, 9+2h) of block kb
//                    are the two Box-Muller outputs of Philox blocks sub = 8 kb + h and 8 kb + 4 + h
//                    (the RNG contract keys a block by the component pair);
//                x_prev[a_p]: two 16-byte loads per k-block from the ancestor's row.
//   C          lane (p, h), register r  ->  output dim 16 cb + h + 4r of particle p; for the
//              Student-t proposal Q xi and G x keep separate accumulators because each component
//              of Q xi is scaled by its own sqrt(nu / chi2) (src/statistics.cc.cpp:385-386, 411).
// An f64 MFMA blocks VALU issue on its SIMD, so this kernel is VALU + MFMA
// serialised: the Philox blocks and Box-Muller pairs (8 per lane per tile at d = 64; ln and
// sincos(2 pi u) from smallops.h) weigh as much as the 128 MFMAs -- it is RNG-bound, not HBM-bound.  Workgroup = 8 waves sharing the LDS factor image and
// an LDS tile counter.
//
// TRIQ: Q is LOWER TRIANGULAR (a Cholesky factor: what cusmc_pf_run_* hands down, and what a caller's Q is
// recognised as when its upper part is zero).  Q Xi then needs the k-blocks kb <= cb only -- 4 NB (NB + 1) / 2
// block-products instead of 4 NB^2, the fragments packed triangular as for the log-density kernels
// (mfma_pack_frags, tri = true).  The triangular instantiations are a translation unit of their own
// (propagate_mfma_tri.hip: this file with CUSMC_TRIQ = 1).
#include "smallops.h"

#ifndef CUSMC_TRIQ
#define CUSMC_TRIQ 0
#endif

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));  // rows are 8-byte aligned in general

__host__ __device__ constexpr int pm_pi(int s, int h) { return 2 * h + (s & 1) + 8 * (s >> 1); }
__host__ __device__ constexpr int pm_q_frags(int nb, bool triq) { return triq ? 2 * nb * (nb + 1) : 4 * nb * nb; }

static __device__ __forceinline__ void pm_normal_pair(const u32x4 r, double &z0, double &z1)
{
  normal_pair(r, z0, z1);  // smallops.h: the same Box-Muller as every other draw path
}

// Every 16 <= d <= 128 and any 8-byte aligned batch.  d <= 96 keeps both factors in LDS (one
// launch); 96 < d <= 128 runs two launches, one factor each (131 KB at d = 128): first
// x = [diag(c)] Q xi, then x += G x_prev[a].
#if !CUSMC_TRIQ
bool propagate_mfma_supported(int d, const void *X_prev, const void *X_out)
{
  return d >= 16 && d <= 128 && (uintptr_t)X_prev % 8 == 0 && (uintptr_t)X_out % 8 == 0;
}
#endif

// MODE 0: initial / R-level draw     x = [diag(c)] Q xi + m0
//      1: propagate, one launch      x = [diag(c)] Q xi + G x_prev[a]
//      2: first half of two          x = [diag(c)] Q xi
//      3: second half                x += G x_prev[a]
//      4: diagonal G, one launch     x = [diag(c)] Q xi + g .* x_prev[a]   (fragsG = the d diagonal entries:
//         a random-walk or AR(1)-per-component state with correlated noise needs no second product; the
//         ancestor's row is gathered straight into the C layout, 8 bytes per lane per output register)
// PAD: d is not 16*NB (factors zero-padded on the host) or rows are not 16-byte aligned: the last
// k-block of the gathered row is loaded element by element, column clamped into the row, columns
// >= d zeroed; normals for pairs past d are not drawn; outputs past d are not stored.
template <int NB, bool MVT, int MODE, bool PAD, bool TRIQ>
__global__ __launch_bounds__(512) void propagate_mfma_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ fragsQ, const double *__restrict__ fragsG, const double *__restrict__ m0,
    int d, double scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t domain, uint32_t first,
    uint32_t count, double *__restrict__ X_out, long num_tiles)
{
  constexpr int D = 16 * NB;
  constexpr int NFRAG = 4 * NB * NB, NFRAGQ = pm_q_frags(NB, TRIQ);
  constexpr bool HAS_Q = MODE != 3, HAS_G = MODE == 1 || MODE == 3, HAS_M0 = MODE == 0, DIAG_G = MODE == 4;
  constexpr bool SPLIT_ACC = (MVT && HAS_Q && HAS_G) || HAS_M0 || MODE == 3;  // accG separate from accQ
  extern __shared__ double lds[];
  double *sQ = lds;                                  // NFRAGQ x 64 (HAS_Q)
  double *sG = sQ + (HAS_Q ? NFRAGQ * 64 : 0);       // NFRAG x 64 (HAS_G), or m0 / diag(G) padded to D
  int *sNext = reinterpret_cast<int *>(sG + (HAS_G ? NFRAG * 64 : (HAS_M0 || DIAG_G ? D : 0)));

  if (HAS_Q)
    for (int i = threadIdx.x; i < NFRAGQ * 32; i += 512) reinterpret_cast<v2d *>(sQ)[i] = reinterpret_cast<const v2d *>(fragsQ)[i];
  if (HAS_G)
    for (int i = threadIdx.x; i < NFRAG * 32; i += 512) reinterpret_cast<v2d *>(sG)[i] = reinterpret_cast<const v2d *>(fragsG)[i];
  if (HAS_M0 && threadIdx.x < D) sG[threadIdx.x] = (int)threadIdx.x < d ? m0[threadIdx.x] : 0.0;
  if (DIAG_G && threadIdx.x < D) sG[threadIdx.x] = (int)threadIdx.x < d ? fragsG[threadIdx.x] : 0.0;
  if (threadIdx.x == 0) *sNext = 0;
  // Student-t: one queue per wave for its open chi^2 draws (smallops.h: ChiQueue), behind the tile counter
  ChiQueue *const chi_q = (MVT && HAS_Q) ? reinterpret_cast<ChiQueue *>(sNext + 2) + (threadIdx.x >> 6) : nullptr;
  if ((MVT && HAS_Q) && (threadIdx.x & 63) == 0) chi_q->count = 0;
  __syncthreads();

  const ChiSquare cs = chi_setup(MVT ? nu : 2.0f);
  const int lane = threadIdx.x & 63;
  const int p = lane & 15, h = lane >> 4;
  const long G_ = gridDim.x;
  const int my_tiles = (int)((num_tiles - 1 - (long)blockIdx.x) / G_) + 1;  // grid <= num_tiles
  auto grab = [&]() -> int {
    int k = 0;
    if (lane == 0) k = atomicAdd(sNext, 1);
    return __builtin_amdgcn_readfirstlane(k);
  };
  int lds_lane = lane;  // opaque per tile: keeps the factor reads as in-loop ds_read_b64
  const int rem = d - 16 * (NB - 1);  // columns of the last k-block that exist (16 unless padded)

  for (int k = grab(); k < my_tiles; k = grab()) {
    asm volatile("" : "+v"(lds_lane));
    const long t = (long)blockIdx.x + (long)k * G_;
    const long local = t * 16 + p;                 // index inside this launch's shard
    const bool live = local < (long)count;
    const uint32_t gi = first + (uint32_t)(live ? local : (long)count - 1);  // global particle

    // ancestor row: two 16-byte loads per k-block (issued first; the RNG below hides them)
    v2d xg[NB][2];
    if constexpr (HAS_G) {
      const uint32_t anc = a ? a[live ? local : (long)count - 1] : gi;
      const double *row = X_prev + (long)anc * d;
#pragma unroll
      for (int kb = 0; kb < (PAD ? NB - 1 : NB); ++kb) {
        xg[kb][0] = *reinterpret_cast<const v2d_a8 *>(row + 16 * kb + 2 * h);
        xg[kb][1] = *reinterpret_cast<const v2d_a8 *>(row + 16 * kb + 8 + 2 * h);
      }
      if constexpr (PAD) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int col = pm_pi(s, h);
          // (a column >= d reads the ancestor's own last element against a zero column of the padded G:
          // no select on the loaded value, which would make the wave wait for the gather before the RNG
          // below had a chance to hide it)
          xg[NB - 1][s >> 1][s & 1] = row[16 * (NB - 1) + (col < rem ? col : rem - 1)];
        }
      }
    }
    // diagonal G: the ancestor's values in the C layout (output dim 16 cb + h + 4 r), issued before the RNG
    double xc[DIAG_G ? NB : 1][4];
    if constexpr (DIAG_G) {
      const uint32_t anc = a ? a[live ? local : (long)count - 1] : gi;
      const double *row = X_prev + (long)anc * d + h;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) xc[cb][r] = (!PAD || 16 * cb + h + 4 * r < d) ? row[16 * cb + 4 * r] : 0.0;
    }
    // normals in operand order: xi[kb][s] = xi_p[16 kb + pi(s, h)]
    double xi[NB][4];
    if constexpr (HAS_Q) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        xi[kb][0] = xi[kb][1] = xi[kb][2] = xi[kb][3] = 0.0;
        if (!PAD || 16 * kb + 2 * h < d)
          pm_normal_pair(philox4x32_10(gi, (uint32_t)(8 * kb + h), step, domain, k0, k1), xi[kb][0], xi[kb][1]);
        if (!PAD || 16 * kb + 8 + 2 * h < d)
          pm_normal_pair(philox4x32_10(gi, (uint32_t)(8 * kb + 4 + h), step, domain, k0, k1), xi[kb][2], xi[kb][3]);
#pragma unroll
        for (int s = 0; s < 4; ++s) xi[kb][s] *= scale;
      }
    }

    v4d accQ[NB], accG[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      accQ[cb] = v4d{0.0, 0.0, 0.0, 0.0};
      if (HAS_M0) {  // C rows are output dims h + 4r of block cb
        const double *b = sG + 16 * cb + h;
        accG[cb] = v4d{b[0], b[4], b[8], b[12]};
      } else {
        accG[cb] = v4d{0.0, 0.0, 0.0, 0.0};
      }
    }
    int f = 0, fq = 0;  // (compile-time after unrolling; TRIQ: the Q image holds the blocks cb >= kb only)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb, ++f) {
          if (HAS_Q && (!TRIQ || cb >= kb)) {
            accQ[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(sQ[fq * 64 + lds_lane], xi[kb][s], accQ[cb], 0, 0, 0);
            ++fq;
          }
          if (HAS_G) {
            v4d &dst = SPLIT_ACC ? accG[cb] : accQ[cb];  // mvn, one launch: one accumulator takes both products
            dst = __builtin_amdgcn_mfma_f64_16x16x4f64(sG[f * 64 + lds_lane], xg[kb][s >> 1][s & 1], dst, 0, 0, 0);
          }
        }
      }
    }
    // x_out[p][16 cb + h + 4 r]
    if (live) {
      double *dst = X_out + local * d + h;
      // the lane's chi^2 draws, four output blocks (16 components: 16 cb + h + 4 r at c = 4 (cb - cb0) + r) at a
      // time: first attempt and squeeze for all of them, then the few still open one per trip
      // (smallops.h: chi_square_batch); groups of four blocks bound the registers the draws hold
      // (fewer draws per batch where registers are short: 4 NB results would spill from d = 64 up)
      constexpr int GB = NB < 4 ? NB : 2;
      double chi[(MVT && HAS_Q) ? 4 * GB : 1];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        if constexpr (MVT && HAS_Q) {
          if (cb % GB == 0) {
            const int cb0 = cb;
            chi_square_clayout<GB>(cs, gi, step, k0, k1, h, [&](int b) { return 16 * (cb0 + b); },
                                   [&](int b, int j) { return cb0 + b < NB && (!PAD || j < d); }, chi, chi_q);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * cb + h + 4 * r;
          if (PAD && j >= d) continue;
          double v = HAS_Q ? accQ[cb][r] : dst[16 * cb + 4 * r];
          if constexpr (MVT && HAS_Q) v *= sqrt((double)nu / chi[4 * (cb % GB) + r]);
          if (SPLIT_ACC) v += accG[cb][r];
          if (DIAG_G) v += fma(sG[j], xc[cb][r], 0.0);  // (the one non-zero term of the dense kernels' sum)
          dst[16 * cb + 4 * r] = v;
        }
      }
    }
  }
}

template <int NB, bool MVT, int MODE, bool PAD, bool TRIQ>
static hipError_t launch_pm(float nu, const double *X_prev, const uint32_t *a, const double *fragsQ,
                            const double *fragsG, const double *m0, int d, double scale, uint64_t seed,
                            uint32_t step, uint32_t domain, uint32_t first, uint32_t count, double *X_out,
                            int num_cus, hipStream_t stream)
{
  constexpr int NFRAG = 4 * NB * NB, NFRAGQ = pm_q_frags(NB, TRIQ);
  constexpr bool HAS_Q = MODE != 3, HAS_G = MODE == 1 || MODE == 3, HAS_M0 = MODE == 0, DIAG_G = MODE == 4;
  const size_t lds_bytes = (size_t)((HAS_Q ? NFRAGQ * 64 : 0) + (HAS_G ? NFRAG * 64 : (HAS_M0 || DIAG_G ? 16 * NB : 0)) + 2) * sizeof(double) +
                           ((MVT && HAS_Q) ? 8 * sizeof(ChiQueue) : 0);
  auto kern = propagate_mfma_kernel<NB, MVT, MODE, PAD, TRIQ>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  const long num_tiles = ((long)count + 15) / 16;
  long blocks = num_cus;
  if (blocks > num_tiles) blocks = num_tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, stream, nu, X_prev, a, fragsQ, fragsG, m0,
                     d, scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out, num_tiles);
  return hipGetLastError();
}

// fragsQ / fragsG: mfma_pack_frags of the factors zero-padded to 16*ceil(d/16) (fragsG == NULL: the
// initial draw, + m0).  g_is_diagonal: fragsG holds the d diagonal entries of G instead.
// launch_propagate_mfma_tri: Q lower triangular, fragsQ packed with tri = true (G dense as before).
#if CUSMC_TRIQ
#define CUSMC_PM_ENTRY launch_propagate_mfma_tri
#else
#define CUSMC_PM_ENTRY launch_propagate_mfma
#endif
hipError_t CUSMC_PM_ENTRY(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *fragsQ, const double *fragsG, bool g_is_diagonal, const double *m0, int d,
                                 double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  const bool mvt = kind == CUSMC_MVT, gather = fragsG != nullptr;
  const bool pad = d % 16 != 0 || (gather && !g_is_diagonal && (uintptr_t)X_prev % 16 != 0);
  constexpr bool TQ = CUSMC_TRIQ != 0;
#define CUSMC_ARGS nu, X_prev, a, fragsQ, fragsG, m0, d, scale, seed, step, domain, first, count, X_out, num_cus, stream
#define CUSMC_PMV(nb, mode)                                                                                  \
  (mvt ? (pad ? launch_pm<nb, true, mode, true, TQ>(CUSMC_ARGS) : launch_pm<nb, true, mode, false, TQ>(CUSMC_ARGS))  \
       : (pad ? launch_pm<nb, false, mode, true, TQ>(CUSMC_ARGS) : launch_pm<nb, false, mode, false, TQ>(CUSMC_ARGS)))
#define CUSMC_PM1(nb) /* both factors fit the LDS */ \
  case nb: return !gather ? CUSMC_PMV(nb, 0) : g_is_diagonal ? CUSMC_PMV(nb, 4) : CUSMC_PMV(nb, 1);
#define CUSMC_PM2(nb) /* one factor per launch */                                   \
  case nb: {                                                                        \
    if (!gather) return CUSMC_PMV(nb, 0);                                           \
    if (g_is_diagonal) return CUSMC_PMV(nb, 4);                                     \
    const hipError_t e = CUSMC_PMV(nb, 2);                                          \
    return e != hipSuccess ? e : CUSMC_PMV(nb, 3);                                  \
  }
#if CUSMC_TRIQ
  // d = 97 .. 112 with a triangular Q: 56 KB + 98 KB fit the LDS together (the Student-t form's queues do not)
  if ((d + 15) / 16 == 7 && gather && !g_is_diagonal && !mvt)
    return pad ? launch_pm<7, false, 1, true, true>(CUSMC_ARGS) : launch_pm<7, false, 1, false, true>(CUSMC_ARGS);
#endif
  switch ((d + 15) / 16) {
    CUSMC_PM1(1) CUSMC_PM1(2) CUSMC_PM1(3) CUSMC_PM1(4) CUSMC_PM1(5) CUSMC_PM1(6)
    CUSMC_PM2(7) CUSMC_PM2(8)
  }
#undef CUSMC_PM_ENTRY
#undef CUSMC_PM1
#undef CUSMC_PM2
#undef CUSMC_PMV
#undef CUSMC_ARGS
  return hipErrorInvalidValue;
}

}  // namespace cusmc
