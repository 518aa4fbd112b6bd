// One bootstrap-filter time step in ONE launch for small state dimension (d <= 8), gfx950.
//
// Replaces, for one t of the reference's time loop (MCMC(), src/mcmc.cpp:292-308),
//     resamplers[resampler](a_t, w_t, N, t, 10)            Sampler::metropolis_hastings, src/samplers.cpp:7-36
//     propagate_K(...)                                     src/mcmc.cpp:90-160
//     reweight_G(...)                                      src/mcmc.cpp:162-237
// i.e. "the weight/resample step of the particle filter".  Given w_{t-1} and x_{t-1} complete,
// everything about particle i of step t is independent of every other particle:
//     a_i  = Metropolis chain i over w_{t-1}                       (B random gathers; from the 4-byte
//                                                                   high-word table once N is large: smallops.h)
//     x_i  = G x_{t-1}[a_i] + [diag(c_i)] Q (scale * xi_i)         (one d-row gather)
//     w_i  = pdf_{0,V}(y_t - F x_i)
// so one lane carries one particle through all three, x_i never leaves registers between the
// proposal and the weight, and the step costs one launch instead of three.
//
// Algorithmic HBM bytes per particle-step: 8d (gathered row) + 8d (x_t)
// + 8 (w_t) + 4 (a_t) = 44 at d = 2; the B weight gathers are served by L2 / Infinity Cache (w is
// 8 MB at N = 1e6).  The kernel is bound by the Philox blocks (B for the chain + d/2 for the
// normals), not by HBM.
//
// Arithmetic is the unfused kernels' (smallops.h; propagate.hip and logpdf_generic.hip for the
// operation order), so a filter run gives the same numbers either way
// (tests/test_gpu_parity.py::test_fused_step_equals_three_launches).
#include "smallops.h"

namespace cusmc {

// SH: the particle array is sharded over devices (cusmc_pf_run_multi_host; the loop of src/mcmc.cpp:292-308 with one
// launch per shard and step): the ancestor's row is read from the shard that owns it -- straight out of a peer's
// HBM --, and w_i and its high word are stored into EVERY shard's copy of the weight vector, so that the step
// needs no gather kernel, no high-word pre-pass and no peer copies behind it.
template <bool SH> struct ShardArg { typedef int type; };  // (the unsharded kernel carries no table: 4 bytes of kernarg)
template <> struct ShardArg<true> { typedef ShardStep type; };

template <int D, bool MVT, bool HI, bool SH>
__global__ __launch_bounds__(256) void pf_step_kernel(
    float nu, const double *__restrict__ w_prev, const uint32_t *__restrict__ w_prev_hi,
    const double *__restrict__ X_prev,
    uint32_t N, uint32_t B, const double *__restrict__ G, const double *__restrict__ Q,
    double scale, int tri, const double *__restrict__ M, const double *__restrict__ shift,
    const double *__restrict__ bias, Epilogue ep, uint32_t k0, uint32_t k1, uint32_t step,
    uint32_t first, uint32_t count, uint32_t *__restrict__ a_out, double *__restrict__ X_out,
    double *__restrict__ w_out, typename ShardArg<SH>::type sh)
{
  __shared__ ChiQueue chi_queues[MVT ? 4 : 1];  // Student-t: a wave's open chi^2 draws, shared out over its lanes
  ChiQueue *const chi_q = MVT ? &chi_queues[threadIdx.x >> 6] : nullptr;
  if (MVT && (threadIdx.x & 63) == 0) chi_q->count = 0;
  if (MVT) chi_wave_fence();
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
    const uint32_t i = first + t;
    // resample
    const uint32_t anc = HI ? metropolis_chain_hi(w_prev, w_prev_hi, N, B, i, step, k0, k1)
                            : metropolis_chain(w_prev, N, B, i, step, k0, k1);
    a_out[t] = anc;
    // propagate (operation order of propagate_kernel)
    double xp[D], xi[D], x[D];
    const double *row = X_prev + (long)anc * D;
    if constexpr (SH) {
      // the owner of row `anc`, by a select chain over ALL kMaxShards entries (the host pads first[] with
      // 0xffffffff): every bound and base pointer is then a kernel argument at a constant index, i.e. a scalar
      // register -- indexed by the lane's own r they are three dependent loads per particle on top of the
      // chain's ten
      uint32_t f = sh.x.first[0];
      row = sh.x.base[0];
#pragma unroll
      for (int s = 1; s < kMaxShards; ++s) {
        const bool ge = anc >= sh.x.first[s];
        row = ge ? sh.x.base[s] : row;
        f = ge ? sh.x.first[s] : f;
      }
      row += (long)(anc - f) * D;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) xp[k] = row[k];
#pragma unroll
    for (int j = 0; j < D; j += 2) {
      double z0, z1;
      normal_pair(philox4x32_10(i, (uint32_t)(j >> 1), step, 2u, k0, k1), z0, z1);
      xi[j] = scale * z0;
      if (j + 1 < D) xi[j + 1] = scale * z1;
    }
    double chi[D];
    if (MVT)  // the particle's D chi^2 draws together (smallops.h: chi_pair_batch)
      chi_square_all<D>(chi_setup(nu), i, step, k0, k1, chi, chi_q);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fma(Q[j * D + k], xi[k], s);
      if (MVT) s = fma(s, sqrt((double)nu / chi[j]), 0.0);  // (an fma, so that no kernel contracts it with the add below)
      double m = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) m = fma(G[j * D + k], xp[k], m);
      x[j] = s + m;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) X_out[(long)t * D + j] = x[j];
    // reweight (operation order of logpdf_generic_kernel)
    double r[D];
#pragma unroll
    for (int k = 0; k < D; ++k) r[k] = x[k] - shift[k];
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int kend = tri ? j + 1 : D;
      double z0 = bias[j], z1 = 0.0;
#pragma unroll
      for (int k = 0; k + 1 < D; k += 2) {
        if (k + 1 < kend) {
          z0 = fma(M[j * D + k], r[k], z0);
          z1 = fma(M[j * D + k + 1], r[k + 1], z1);
        } else if (k < kend) {
          z0 = fma(M[j * D + k], r[k], z0);
        }
      }
      if ((D & 1) && D - 1 < kend) z0 = fma(M[j * D + D - 1], r[D - 1], z0);
      const double z = z0 + z1;
      q = fma(z, z, q);
    }
    const double wi = finish_generic(q, ep);
    w_out[t] = wi;
    if constexpr (SH) {
      const uint32_t hi = (uint32_t)(__builtin_bit_cast(uint64_t, wi) >> 32);  // (resample.hip: hiword_kernel)
      for (int r = 0; r < sh.x.n; ++r) {
        sh.w_dst[r][i] = wi;
        sh.whi_dst[r][i] = hi;
      }
    }
  }
}

bool pf_step_supported(int d) { return d >= 1 && d <= 8; }

hipError_t launch_pf_step(int kind, float nu, const double *w_prev, const uint32_t *w_prev_hi,
                          const double *X_prev,
                          uint32_t N, int d, uint32_t B, const double *G, const double *Q,
                          double scale, bool tri, const double *M, const double *shift,
                          const double *bias, const Epilogue &ep, uint64_t seed, uint32_t step,
                          uint32_t first, uint32_t count, uint32_t *a_out, double *X_out,
                          double *w_out, int num_cus, hipStream_t stream, const ShardStep *sharded)
{
  if (count == 0) return hipSuccess;
  long blocks = ((long)count + 255) / 256;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
#define CUSMC_PFK(D, SH)                                                                                            \
  (kind == CUSMC_MVT ? (w_prev_hi ? pf_step_kernel<D, true, true, SH> : pf_step_kernel<D, true, false, SH>)         \
                     : (w_prev_hi ? pf_step_kernel<D, false, true, SH> : pf_step_kernel<D, false, false, SH>))
#define CUSMC_PFL(D, SH, arg)                                                                              \
  hipLaunchKernelGGL(CUSMC_PFK(D, SH), dim3((unsigned)blocks), dim3(256), 0, stream, nu, w_prev, w_prev_hi,    \
                     X_prev, N, B, G, Q, scale, (int)tri, M, shift, bias, ep, (uint32_t)seed,                  \
                     (uint32_t)(seed >> 32), step, first, count, a_out, X_out, w_out, arg)
#define CUSMC_PF(D)                         \
  case D: {                                 \
    if (sharded) CUSMC_PFL(D, true, *sharded); \
    else CUSMC_PFL(D, false, 0);            \
    break;                                  \
  }
  switch (d) {
    CUSMC_PF(1) CUSMC_PF(2) CUSMC_PF(3) CUSMC_PF(4) CUSMC_PF(5) CUSMC_PF(6) CUSMC_PF(7) CUSMC_PF(8)
    default: return hipErrorInvalidValue;
  }
#undef CUSMC_PF
#undef CUSMC_PFK
#undef CUSMC_PFL
  return hipGetLastError();
}

}  // namespace cusmc
