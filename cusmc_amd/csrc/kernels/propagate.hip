// Proposal draws on gfx950: gather + RNG + two mat-vecs in ONE launch.
//
// Replaces the reference's four-launch propagate pipeline (mvn_sample_setup_kernel ->
// mvn_sample_norm_rand_kernel -> mvn_sample_Gmu_kernel -> mvn_sample_kernel,
// src/mvn_dist.cu.cpp:15-172, host wrapper :175-319 with its AoS<->flat repacking and ten
// cudaMallocs per call; the mvt twins src/mvt_dist.cu.cpp:63-223) and the CPU loop of propagate_K
// (src/mcmc.cpp:116-140 -> MultiVariate*Distribution::sample, src/statistics.cc.cpp:224-259,
// :355-412), and with G == NULL the initial draw of initialize() (src/mcmc.cpp:76-82):
//
//     x_out[i] = [diag(c_i)] Q (scale * xi_i) + ( G ? G x_prev[a_i] : m0 )
//
// xi_i ~ N(0, I) by Box-Muller on Philox blocks (philox.h: no RNG state array -- the reference
// keeps 48 bytes of curandState per ELEMENT), c_ij = sqrt(nu / chi2_nu) per component for the
// Student-t proposal exactly as the reference scales it (src/statistics.cc.cpp:385-386,411;
// SURVEY.md F7), chi2 by Marsaglia-Tsang as in src/mvt_dist.cu.cpp:20-61 with a deterministic
// counter advance.
//
// Mapping: lane = particle.  The ancestor rows are gathered into LDS with row-coalesced loads,
// each lane writes its own normals into a second LDS row, then walks the d outputs with Q and G
// read through wave-uniform (scalar) loads.  Rows are padded to an odd stride (conflict-free
// ds_read_b64).  This is the general-d first version (d <= 159, limited by two LDS rows per
// particle); DESIGN.md lists the MFMA formulation as the next step for d >= 32.
#include <hip/hip_runtime.h>

#include "smallops.h"

namespace cusmc {

typedef double v2d __attribute__((ext_vector_type(2)));

template <int T>
__global__ __launch_bounds__(T) void propagate_kernel(
    int kind, float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ G, const double *__restrict__ Q, const double *__restrict__ m0,
    int d, double scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t domain, uint32_t first,
    uint32_t count, double *__restrict__ X_out, long num_tiles)
{
  extern __shared__ double lds[];
  const int stride = d | 1;
  double *sX = lds;                // gathered ancestor rows
  double *sXi = lds + T * stride;  // this particle's normals

  for (long tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const long base = tile * T;
    const long rows = ((long)count - base) < T ? ((long)count - base) : T;
    __syncthreads();
    if (G) {
      int row = (int)(threadIdx.x / d), col = (int)(threadIdx.x % d);
      const int drow = T / d, dcol = T % d;
      for (long e = threadIdx.x; e < rows * d; e += T) {
        const uint32_t anc = a ? a[base + row] : first + (uint32_t)(base + row);
        sX[row * stride + col] = X_prev[(long)anc * d + col];
        row += drow;
        col += dcol;
        if (col >= d) { col -= d; ++row; }
      }
    }
    const uint32_t i = first + (uint32_t)(base + threadIdx.x);
    double *xi = sXi + threadIdx.x * stride;
    if ((long)threadIdx.x < rows) {
      for (int j = 0; j < d; j += 2) {
        double z0, z1;
        normal_pair(philox4x32_10(i, (uint32_t)(j >> 1), step, domain, k0, k1), z0, z1);
        xi[j] = scale * z0;
        if (j + 1 < d) xi[j + 1] = scale * z1;
      }
    }
    __syncthreads();
    if ((long)threadIdx.x < rows) {
      const double *xp = sX + threadIdx.x * stride;
      double *dst = X_out + (base + threadIdx.x) * d;
      for (int j = 0; j < d; ++j) {
        const double *Qj = Q + (long)j * d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) s = fma(Qj[k], xi[k], s);
        if (kind == CUSMC_MVT) s = fma(s, sqrt((double)nu / chi_square_for(i, (uint32_t)j, step, k0, k1, nu)), 0.0);  // (an fma, so that no kernel contracts it with the add below)
        double m;
        if (G) {
          const double *Gj = G + (long)j * d;
          m = 0.0;
          for (int k = 0; k < d; ++k) m = fma(Gj[k], xp[k], m);
        } else {
          m = m0[j];
        }
        dst[j] = s + m;
      }
    }
  }
}

// Diagonal G and Q (random-walk and independent-component models: the reference's own
// generateInput() uses G = I, W = 0.001 I, src/mcmc.cpp:22-23): the two mat-vecs collapse to one
// multiply per component, so there is nothing to stage and nothing for the matrix cores -- one lane
// per (particle, component pair), i.e. per Philox block, consecutive lanes on consecutive pairs of
// one row (16-byte segments, fully coalesced), any d.  Same operations as the general kernels
// (fma(q, xi, 0) is the only non-zero term of their sum), hence the same values.
// rows a lane of the Normal kernel takes per trip (8: 325 us against 302 for 1e6 x 64; forcing 6 or 8 waves per SIMD
// spills: 540 .. 830 us)
constexpr int kDiagRows = 4;

// CHI: 0 Normal; 1, 2 Student-t with the closed-form chi^2 of nu = 2, 4 (one more Philox block and two ln per pair, no
// rejection -- same loop as the Normal draw); 4: the closed form of any other integer nu <= 16 (RNG contract 4:
// ceil(nu / 4) blocks of uniforms, a Box-Muller pair for odd nu), same loop; 3 Student-t by Marsaglia-Tsang (any other
// nu), PPL pairs of a row per lane and trip
template <int CHI, int PPL = 4>
__global__ __launch_bounds__(256) void propagate_diag_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ gdiag, const double *__restrict__ qdiag, const double *__restrict__ m0,
    int d, int pw_log2, double scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t domain,
    uint32_t first, uint32_t count, double *__restrict__ X_out)
{
  constexpr bool MVT = CHI != 0;
  // 2^pw_log2 lanes per row (>= the number of pairs up to 256, no division anywhere); a block
  // covers 256 >> pw_log2 rows per pass
  const int pairs = (d + 1) / 2;
  const int pw = 1 << pw_log2;
  const int rows_per_block = 256 >> pw_log2;
  const int lane_pr = (int)threadIdx.x & (pw - 1);
  const uint32_t lane_row = threadIdx.x >> pw_log2;
  const ChiSquare cs = chi_setup(MVT ? nu : 2.0f);
  if constexpr (CHI != 3) {
    // A lane takes its pair of U = 4 rows at a time -- the four ancestor indices, then the four 16-byte
    // pieces of the ancestors' rows are requested before the first normal is drawn.  One row at a time, a CU's 32
    // waves kept 32 KB in flight and the kernel ran at what that buys (3.8 TB/s of gather + store, RNG on top:
    // 375 us for 1e6 x 64; 302 us now, which is the VALU time of the draws: ~1400 cycles per wave and pair);
    // the values are the same operations as before.
    constexpr int U = kDiagRows;
    const uint32_t stride = gridDim.x * rows_per_block;
    const bool wide = (d & 1) == 0 && (((uintptr_t)X_prev | (uintptr_t)X_out) & 15) == 0;  // 16-byte pieces
    for (uint32_t il0 = blockIdx.x * rows_per_block + lane_row; il0 < count; il0 += U * stride) {
      uint32_t il[U];
      long anc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        il[u] = il0 + u * stride;
        const uint32_t ic = il[u] < count ? il[u] : count - 1;  // (clamped: loaded, never stored)
        anc[u] = gdiag ? (a ? (long)a[ic] : (long)(first + ic)) : 0;
      }
      for (int pr = lane_pr; pr < pairs; pr += pw) {
        const int j = 2 * pr;
        const bool both = j + 1 < d;
        double x[U][2];
        if (gdiag) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double *src = X_prev + anc[u] * d + j;
            if (wide) {
              const v2d v = *reinterpret_cast<const v2d *>(src);
              x[u][0] = v[0], x[u][1] = v[1];
            } else {
              x[u][0] = src[0];
              x[u][1] = both ? src[1] : 0.0;
            }
          }
        }
        const double q0 = qdiag[j], q1 = both ? qdiag[j + 1] : 0.0;
        const double t0 = gdiag ? gdiag[j] : m0[j], t1 = both ? (gdiag ? gdiag[j + 1] : m0[j + 1]) : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (il[u] >= count) continue;
          double z0, z1;
          normal_pair(philox4x32_10(first + il[u], (uint32_t)pr, step, domain, k0, k1), z0, z1);
          double s0 = fma(q0, scale * z0, 0.0), s1 = fma(q1, scale * z1, 0.0);
          if constexpr (MVT) {
            double c0, c1;
            if constexpr (CHI == 4) chi_closed_pair_any(cs, first + il[u], (uint32_t)pr, step, k0, k1, c0, c1);
            else chi_closed_pair<CHI>(first + il[u], (uint32_t)pr, step, k0, k1, c0, c1);
            s0 = fma(s0, sqrt((double)nu / c0), 0.0);  // (an fma, so that no kernel contracts it with the add below)
            s1 = fma(s1, sqrt((double)nu / c1), 0.0);
          }
          const double o0 = s0 + (gdiag ? fma(t0, x[u][0], 0.0) : t0);
          const double o1 = s1 + (gdiag ? fma(t1, x[u][1], 0.0) : t1);
          double *dst = X_out + (long)il[u] * d + j;
          if (wide) {
            *reinterpret_cast<v2d *>(dst) = v2d{o0, o1};
          } else {
            dst[0] = o0;
            if (both) dst[1] = o1;
          }
        }
      }
    }
    return;
  }
  __shared__ ChiQueue chi_queues[4];  // one per wave: its open chi^2 draws, shared out over its lanes (smallops.h)
  ChiQueue *const chi_q = &chi_queues[threadIdx.x >> 6];
  if ((threadIdx.x & 63) == 0) chi_q->count = 0;
  chi_wave_fence();
  for (uint32_t il = blockIdx.x * rows_per_block + lane_row; il < count; il += gridDim.x * rows_per_block) {
    const uint32_t i = first + il;
    const long anc = gdiag ? (a ? (long)a[il] : (long)i) : 0;
    {
      // Student-t by rejection: a lane takes PPL = 4 pairs of its row at a time (the launcher gives a row a quarter
      // of the lanes; 1 below d = 7), so that their eight chi^2 draws go through chi_pair_batch together -- with two
      // draws per batch the wave pays the slow path of nearly every batch (smallops.h): 1400 -> 1165 us for
      // 1e6 x 64 under contract 1; the pairs' values, same operations as above
      for (int pr0 = lane_pr; pr0 < pairs; pr0 += PPL * pw) {
        double z[PPL][2];
#pragma unroll
        for (int q = 0; q < PPL; ++q) z[q][0] = z[q][1] = 0.0;
#pragma unroll 1
        for (int q = 0; q < PPL; ++q) {
          double z0 = 0.0, z1 = 0.0;
          if (pr0 + q * pw < pairs) normal_pair(philox4x32_10(i, (uint32_t)(pr0 + q * pw), step, domain, k0, k1), z0, z1);
#pragma unroll
          for (int qq = 0; qq < PPL; ++qq) {
            z[qq][0] = qq == q ? z0 : z[qq][0];
            z[qq][1] = qq == q ? z1 : z[qq][1];
          }
        }
        double chi[2 * PPL];
        chi_pair_batch<PPL>(cs, i, step, k0, k1, [&](int c) { return pr0 + c * pw; }, [&](int c) { return pr0 + c * pw < pairs; }, chi, chi_q);
#pragma unroll
        for (int q = 0; q < PPL; ++q)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int j = 2 * (pr0 + q * pw) + c;
            if (j < d) {
              double s = fma(qdiag[j], scale * z[q][c], 0.0);
              s = fma(s, sqrt((double)nu / chi[2 * q + c]), 0.0);  // (an fma, so that no kernel contracts it with the add below)
              const double m = gdiag ? fma(gdiag[j], X_prev[anc * d + j], 0.0) : m0[j];
              X_out[(long)il * d + j] = s + m;
            }
          }
      }
    }
  }
}

hipError_t launch_propagate_diag(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *gdiag, const double *qdiag, const double *m0, int d,
                                 double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  const int pairs = (d + 1) / 2;
  int pw_log2 = 0;
  while ((1 << pw_log2) < pairs && pw_log2 < 8) ++pw_log2;
  const int chi = kind != CUSMC_MVT ? 0 : nu == 2.0f ? 1 : nu == 4.0f ? 2 : chi_nu_closed(nu) ? 4 : 3;  // (smallops.h: chi_setup)
  const bool four = chi == 3 && pairs >= 4;
  if (four) pw_log2 -= 2;  // four pairs per lane: eight chi^2 draws per batch
  const long rows_per_block = (long)(256 >> pw_log2) * (chi != 3 ? kDiagRows : 1);  // (four rows per lane and trip)
  long blocks = ((long)count + rows_per_block - 1) / rows_per_block;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  auto kern = chi == 0 ? propagate_diag_kernel<0> : chi == 1 ? propagate_diag_kernel<1> : chi == 2 ? propagate_diag_kernel<2>
              : chi == 4 ? propagate_diag_kernel<4> : four ? propagate_diag_kernel<3, 4> : propagate_diag_kernel<3, 1>;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, nu, X_prev, a, gdiag, qdiag, m0, d,
                     pw_log2, scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out);
  return hipGetLastError();
}

// Dense G / Q above d = 128, where the factors no longer fit the LDS of the matrix-core kernel: one
// WORKGROUP per particle, thread j owns output component j.  The normals and the ancestor's row go
// through LDS; QT / GT are the TRANSPOSED factors, so a wavefront reads one contiguous row segment
// per k (coalesced, L2-resident).  Sums run over k in order, like the lane-per-particle kernel.
// A single R-level draw at d = 256 costs one small launch here instead of one lane walking 65k FMAs.
template <bool MVT>
__global__ __launch_bounds__(256) void propagate_row_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ GT, const double *__restrict__ QT, const double *__restrict__ m0,
    int d, double scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t domain, uint32_t first,
    uint32_t count, double *__restrict__ X_out)
{
  extern __shared__ double lds[];
  double *sXi = lds, *sXp = lds + d + (d & 1);
  for (uint32_t il = blockIdx.x; il < count; il += gridDim.x) {
    const uint32_t i = first + il;
    __syncthreads();
    for (int pr = threadIdx.x; 2 * pr < d; pr += 256) {
      double z0, z1;
      normal_pair(philox4x32_10(i, (uint32_t)pr, step, domain, k0, k1), z0, z1);
      sXi[2 * pr] = scale * z0;
      if (2 * pr + 1 < d) sXi[2 * pr + 1] = scale * z1;
    }
    if (GT) {
      const long anc = a ? (long)a[il] : (long)i;
      for (int k = threadIdx.x; k < d; k += 256) sXp[k] = X_prev[anc * d + k];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 256) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s = fma(QT[(long)k * d + j], sXi[k], s);
      if (MVT) s = fma(s, sqrt((double)nu / chi_square_for(i, (uint32_t)j, step, k0, k1, nu)), 0.0);
      double m;
      if (GT) {
        m = 0.0;
        for (int k = 0; k < d; ++k) m = fma(GT[(long)k * d + j], sXp[k], m);
      } else {
        m = m0[j];
      }
      X_out[(long)il * d + j] = s + m;
    }
  }
}

hipError_t launch_propagate_rows(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *GT, const double *QT, const double *m0, int d,
                                 double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  long blocks = count;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  const size_t lds_bytes = (size_t)2 * (d + (d & 1)) * sizeof(double);
  auto kern = kind == CUSMC_MVT ? propagate_row_kernel<true> : propagate_row_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, nu, X_prev, a, GT, QT, m0, d,
                     scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out);
  return hipGetLastError();
}

// Dense G / Q below d = 16: one lane per particle with everything in registers (the row gather is
// D loads, the factors come through wave-uniform scalar loads).  Operation order of propagate_kernel
// and pf_step.hip.
template <int D, bool MVT>
__global__ __launch_bounds__(256) void propagate_small_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ G, const double *__restrict__ Q, const double *__restrict__ m0,
    double scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t domain, uint32_t first,
    uint32_t count, double *__restrict__ X_out)
{
  __shared__ ChiQueue chi_queues[MVT ? 4 : 1];  // Student-t: a wave's open chi^2 draws, shared out over its lanes
  ChiQueue *const chi_q = MVT ? &chi_queues[threadIdx.x >> 6] : nullptr;
  if (MVT && (threadIdx.x & 63) == 0) chi_q->count = 0;
  if (MVT) chi_wave_fence();
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < count; t += stride) {
    const uint32_t i = first + t;
    double xp[D], xi[D];
    if (G) {
      const long anc = a ? (long)a[t] : (long)i;
#pragma unroll
      for (int k = 0; k < D; ++k) xp[k] = X_prev[anc * D + k];
    }
#pragma unroll
    for (int j = 0; j < D; j += 2) {
      double z0, z1;
      normal_pair(philox4x32_10(i, (uint32_t)(j >> 1), step, domain, k0, k1), z0, z1);
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
      if (MVT) s = fma(s, sqrt((double)nu / chi[j]), 0.0);
      double m;
      if (G) {
        m = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) m = fma(G[j * D + k], xp[k], m);
      } else {
        m = m0[j];
      }
      X_out[(long)t * D + j] = s + m;
    }
  }
}

template <int D>
static hipError_t launch_small(int kind, float nu, const double *X_prev, const uint32_t *a, const double *G,
                               const double *Q, const double *m0, double scale, uint64_t seed, uint32_t step,
                               uint32_t domain, uint32_t first, uint32_t count, double *X_out, int num_cus,
                               hipStream_t stream)
{
  long blocks = ((long)count + 255) / 256;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  auto kern = kind == CUSMC_MVT ? propagate_small_kernel<D, true> : propagate_small_kernel<D, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, nu, X_prev, a, G, Q, m0, scale,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out);
  return hipGetLastError();
}

template <int T>
static hipError_t launch_t(int kind, float nu, const double *X_prev, const uint32_t *a,
                           const double *G, const double *Q, const double *m0, int d,
                           double scale, uint64_t seed, uint32_t step, uint32_t domain,
                           uint32_t first, uint32_t count, double *X_out, int num_cus,
                           hipStream_t stream)
{
  const size_t lds_bytes = (size_t)2 * T * (d | 1) * sizeof(double);
  auto kern = propagate_kernel<T>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  const long num_tiles = ((long)count + T - 1) / T;
  int per_cu = (int)((160 * 1024) / lds_bytes);
  const int max_per_cu = 2048 / T > 8 ? 8 : 2048 / T;
  per_cu = per_cu > max_per_cu ? max_per_cu : (per_cu < 1 ? 1 : per_cu);
  long blocks = (long)num_cus * per_cu;
  if (blocks > num_tiles) blocks = num_tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(T), lds_bytes, stream, kind, nu, X_prev, a,
                     G, Q, m0, d, scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, domain,
                     first, count, X_out, num_tiles);
  return hipGetLastError();
}

hipError_t launch_propagate(int kind, float nu, const double *X_prev, const uint32_t *a,
                            const double *G, const double *Q, const double *m0, int d,
                            double scale, uint64_t seed, uint32_t step, uint32_t domain,
                            uint32_t first, uint32_t count, double *X_out, int num_cus,
                            hipStream_t stream)
{
  if (count == 0) return hipSuccess;
#define CUSMC_SMALL(D) \
  case D: return launch_small<D>(kind, nu, X_prev, a, G, Q, m0, scale, seed, step, domain, first, count, X_out, num_cus, stream);
  switch (d) {
    CUSMC_SMALL(1) CUSMC_SMALL(2) CUSMC_SMALL(3) CUSMC_SMALL(4) CUSMC_SMALL(5) CUSMC_SMALL(6) CUSMC_SMALL(7)
    CUSMC_SMALL(8) CUSMC_SMALL(9) CUSMC_SMALL(10) CUSMC_SMALL(11) CUSMC_SMALL(12) CUSMC_SMALL(13)
    CUSMC_SMALL(14) CUSMC_SMALL(15)
    default: break;
  }
#undef CUSMC_SMALL
  const size_t row_bytes = (size_t)2 * (d | 1) * 8;
  if (64 * row_bytes > 160 * 1024) return hipErrorInvalidValue;
  if (256 * row_bytes <= 64 * 1024)
    return launch_t<256>(kind, nu, X_prev, a, G, Q, m0, d, scale, seed, step, domain, first, count,
                         X_out, num_cus, stream);
  if (128 * row_bytes <= 64 * 1024)
    return launch_t<128>(kind, nu, X_prev, a, G, Q, m0, d, scale, seed, step, domain, first, count,
                         X_out, num_cus, stream);
  return launch_t<64>(kind, nu, X_prev, a, G, Q, m0, d, scale, seed, step, domain, first, count,
                      X_out, num_cus, stream);
}

}  // namespace cusmc
