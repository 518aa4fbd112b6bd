// Proposal draws for d = 16*NB <= 64 on the f64 matrix cores: gather + RNG + two dense mat-vecs
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
#include "smallops.h"

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int pm_pi(int s, int h) { return 2 * h + (s & 1) + 8 * (s >> 1); }

static __device__ __forceinline__ void pm_normal_pair(const u32x4 r, double &z0, double &z1)
{
  normal_pair(r, z0, z1);  // smallops.h: the same Box-Muller as every other draw path
}

// chi^2_nu = 2 Gamma(nu/2, 1); counter layout as oracle/cusmc_oracle.c:chi_square_for.
static __device__ __attribute__((noinline)) double pm_chi_square(uint32_t particle, uint32_t j, uint32_t step, uint32_t k0, uint32_t k1,
                                       float nu)
{
  double a = 0.5 * (double)nu;
  double boost = 1.0;
  if (a < 1.0) {
    const u32x4 r = philox4x32_10(particle, j * 64u + 63u, step, 5u, k0, k1);
    boost = pow(1.0 - u01_53(r.x, r.y), 1.0 / a);
    a += 1.0;
  }
  const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
  double g = dd;
  for (uint32_t m = 0; m < 63u; ++m) {
    double z0, z1;
    pm_normal_pair(philox4x32_10(particle, j * 64u + m, step, 3u, k0, k1), z0, z1);
    double v = 1.0 + c * z0;
    if (v <= 0.0) continue;
    v = v * v * v;
    const u32x4 r = philox4x32_10(particle, j * 64u + m, step, 5u, k0, k1);
    const double u = 1.0 - u01_53(r.x, r.y);
    if (ln_pos(u) < 0.5 * z0 * z0 + dd - dd * v + dd * ln_pos(v)) {
      g = dd * v;
      break;
    }
  }
  return 2.0 * g * boost;
}

bool propagate_mfma_supported(int d, const void *X_prev, const void *X_out)
{
  if (d % 16 != 0 || d > 64) return false;
  return ((uintptr_t)X_prev % 16 == 0) && ((uintptr_t)X_out % 8 == 0);
}

// GATHER = true: propagate (G x_prev[a]); false: initial / R-level draw (+ m0).
template <int NB, bool MVT, bool GATHER>
__global__ __launch_bounds__(512) void propagate_mfma_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ fragsQ, const double *__restrict__ fragsG, const double *__restrict__ m0,
    double scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t domain, uint32_t first, uint32_t count,
    double *__restrict__ X_out, long num_tiles)
{
  constexpr int D = 16 * NB;
  constexpr int NFRAG = 4 * NB * NB;
  extern __shared__ double lds[];
  double *sQ = lds;                                  // NFRAG x 64
  double *sG = sQ + NFRAG * 64;                      // NFRAG x 64 (GATHER) or m0 (D doubles)
  int *sNext = reinterpret_cast<int *>(sG + (GATHER ? NFRAG * 64 : D));

  for (int i = threadIdx.x; i < NFRAG * 32; i += 512) {
    reinterpret_cast<v2d *>(sQ)[i] = reinterpret_cast<const v2d *>(fragsQ)[i];
    if (GATHER) reinterpret_cast<v2d *>(sG)[i] = reinterpret_cast<const v2d *>(fragsG)[i];
  }
  if (!GATHER && threadIdx.x < D) sG[threadIdx.x] = m0[threadIdx.x];
  if (threadIdx.x == 0) *sNext = 0;
  __syncthreads();

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

  for (int k = grab(); k < my_tiles; k = grab()) {
    asm volatile("" : "+v"(lds_lane));
    const long t = (long)blockIdx.x + (long)k * G_;
    const long local = t * 16 + p;                 // index inside this launch's shard
    const bool live = local < (long)count;
    const uint32_t gi = first + (uint32_t)(live ? local : (long)count - 1);  // global particle

    // ancestor row: two 16-byte loads per k-block (issued first; the RNG below hides them)
    v2d xg[NB][2];
    if (GATHER) {
      const uint32_t anc = a ? a[live ? local : (long)count - 1] : gi;
      const double *src = X_prev + (long)anc * D + 2 * h;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        xg[kb][0] = *reinterpret_cast<const v2d *>(src + 16 * kb);
        xg[kb][1] = *reinterpret_cast<const v2d *>(src + 16 * kb + 8);
      }
    }
    // normals in operand order: xi[kb][s] = xi_p[16 kb + pi(s, h)]
    double xi[NB][4];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      pm_normal_pair(philox4x32_10(gi, (uint32_t)(8 * kb + h), step, domain, k0, k1), xi[kb][0], xi[kb][1]);
      pm_normal_pair(philox4x32_10(gi, (uint32_t)(8 * kb + 4 + h), step, domain, k0, k1), xi[kb][2], xi[kb][3]);
#pragma unroll
      for (int s = 0; s < 4; ++s) xi[kb][s] *= scale;
    }

    v4d accQ[NB], accG[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      accQ[cb] = v4d{0.0, 0.0, 0.0, 0.0};
      if (GATHER) {
        accG[cb] = v4d{0.0, 0.0, 0.0, 0.0};
      } else {  // C rows are output dims h + 4r of block cb
        const double *b = sG + 16 * cb + h;
        accG[cb] = v4d{b[0], b[4], b[8], b[12]};
      }
    }
    int f = 0;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb, ++f) {
          accQ[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(sQ[f * 64 + lds_lane], xi[kb][s], accQ[cb], 0, 0, 0);
          if (GATHER) {
            v4d &dst = MVT ? accG[cb] : accQ[cb];  // mvn: one accumulator takes both products
            dst = __builtin_amdgcn_mfma_f64_16x16x4f64(sG[f * 64 + lds_lane], xg[kb][s >> 1][s & 1], dst, 0, 0, 0);
          }
        }
      }
    }
    // x_out[p][16 cb + h + 4 r]
    if (live) {
      double *dst = X_out + local * D + h;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = accQ[cb][r];
          if (MVT) {
            const uint32_t j = (uint32_t)(16 * cb + h + 4 * r);
            v *= sqrt((double)nu / pm_chi_square(gi, j, step, k0, k1, nu));
          }
          if (MVT || !GATHER) v += accG[cb][r];
          dst[16 * cb + 4 * r] = v;
        }
      }
    }
  }
}

template <int NB, bool MVT, bool GATHER>
static hipError_t launch_pm(float nu, const double *X_prev, const uint32_t *a, const double *fragsQ,
                            const double *fragsG, const double *m0, double scale, uint64_t seed, uint32_t step,
                            uint32_t domain, uint32_t first, uint32_t count, double *X_out, int num_cus,
                            hipStream_t stream)
{
  constexpr int NFRAG = 4 * NB * NB;
  const size_t lds_bytes = (size_t)(NFRAG * 64 + (GATHER ? NFRAG * 64 : 16 * NB) + 2) * sizeof(double);
  auto kern = propagate_mfma_kernel<NB, MVT, GATHER>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  const long num_tiles = ((long)count + 15) / 16;
  long blocks = num_cus;
  if (blocks > num_tiles) blocks = num_tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, stream, nu, X_prev, a, fragsQ, fragsG, m0,
                     scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out, num_tiles);
  return hipGetLastError();
}

hipError_t launch_propagate_mfma(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *fragsQ, const double *fragsG, const double *m0, int d,
                                 double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  const bool mvt = kind == CUSMC_MVT, gather = fragsG != nullptr;
#define CUSMC_PM(nb)                                                                                        \
  case nb:                                                                                                  \
    if (gather)                                                                                             \
      return mvt ? launch_pm<nb, true, true>(nu, X_prev, a, fragsQ, fragsG, m0, scale, seed, step, domain,  \
                                             first, count, X_out, num_cus, stream)                          \
                 : launch_pm<nb, false, true>(nu, X_prev, a, fragsQ, fragsG, m0, scale, seed, step, domain, \
                                              first, count, X_out, num_cus, stream);                        \
    return mvt ? launch_pm<nb, true, false>(nu, X_prev, a, fragsQ, fragsG, m0, scale, seed, step, domain,   \
                                            first, count, X_out, num_cus, stream)                           \
               : launch_pm<nb, false, false>(nu, X_prev, a, fragsQ, fragsG, m0, scale, seed, step, domain,  \
                                             first, count, X_out, num_cus, stream);
  switch (d / 16) {
    CUSMC_PM(1)
    CUSMC_PM(2)
    CUSMC_PM(3)
    CUSMC_PM(4)
  }
#undef CUSMC_PM
  return hipErrorInvalidValue;
}

}  // namespace cusmc
