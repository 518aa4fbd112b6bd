// Per-lane building blocks shared by the lane = particle kernels (resample.hip, propagate.hip,
// logpdf_generic.hip) and by the fused filter step (pf_step.hip), so that the fused and the
// unfused paths execute the same arithmetic in the same order.  Contract: philox.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../launch.h"
#include "../philox.h"
#include "../../../include/cusmc_hip.h"

namespace cusmc {

// Box-Muller on one Philox block: u1 in (0,1], u2 in [0,1).
static __device__ __forceinline__ void normal_pair(const u32x4 r, double &z0, double &z1)
{
  const double u1 = 1.0 - u01_53(r.x, r.y);  // (0,1]
  const double u2 = u01_53(r.z, r.w);        // [0,1)
  const double rad = sqrt(-2.0 * log(u1));
  const double ang = 2.0 * 3.14159265358979323846 * u2;
  z0 = rad * cos(ang);
  z1 = rad * sin(ang);
}

// chi^2_nu = 2 Gamma(nu/2, 1) by Marsaglia-Tsang (the reference's device sampler,
// src/mvt_dist.cu.cpp:20-61); counter layout as oracle/cusmc_oracle.c:chi_square_for.
static __device__ double chi_square_for(uint32_t particle, uint32_t j, uint32_t step, uint32_t k0,
                                        uint32_t k1, float nu)
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
    normal_pair(philox4x32_10(particle, j * 64u + m, step, 3u, k0, k1), z0, z1);
    double v = 1.0 + c * z0;
    if (v <= 0.0) continue;
    v = v * v * v;
    const u32x4 r = philox4x32_10(particle, j * 64u + m, step, 5u, k0, k1);
    const double u = 1.0 - u01_53(r.x, r.y);
    if (log(u) < 0.5 * z0 * z0 + dd - dd * v + dd * log(v)) {
      g = dd * v;
      break;
    }
  }
  return 2.0 * g * boost;
}

// One Metropolis chain (Sampler::metropolis_hastings, src/samplers.cpp:21-35):
//     k = i;  B times { u ~ U[0,1); j ~ UnifInt[0,N); if (u <= w[j] / w[k]) k = j; }
// The draw order (u, then j), the division and the `<=` are the reference's, so a NaN ratio never
// accepts.  The random numbers and the gather of step n do not depend on the chain state, only
// the compare does, so the loop is unrolled by four: four Philox blocks and four gathers are in
// flight before the four dependent accept tests.
static __device__ __forceinline__ uint32_t metropolis_chain(const double *__restrict__ w, uint32_t N,
                                                            uint32_t B, uint32_t i, uint32_t step,
                                                            uint32_t k0, uint32_t k1)
{
  uint32_t k = i;
  double wk = w[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    double u[4], wj[4];
    uint32_t j[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 r = philox4x32_10(i, n + c, step, 1u, k0, k1);
      u[c] = u01_53(r.x, r.y);
      j[c] = uint_below(r.z, r.w, N);
      wj[c] = w[j[c]];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (u[c] <= wj[c] / wk) {
        k = j[c];
        wk = wj[c];
      }
    }
  }
  for (; n < B; ++n) {
    const u32x4 r = philox4x32_10(i, n, step, 1u, k0, k1);
    const double u = u01_53(r.x, r.y);
    const uint32_t j = uint_below(r.z, r.w, N);
    const double wj = w[j];
    if (u <= wj / wk) {
      k = j;
      wk = wj;
    }
  }
  return k;
}

static __device__ __forceinline__ double finish_generic(double q, const Epilogue &ep)
{
  double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p(q * ep.inv_nu)
                                     : ep.lognorm - 0.5 * q;
  return ep.out_density ? exp(lp) : lp;
}

}  // namespace cusmc
