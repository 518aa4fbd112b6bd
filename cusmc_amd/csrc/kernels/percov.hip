// Per-particle covariances, d <= 16: batched small-matrix Cholesky and the log-density that goes
// with it.  The reference has ONE covariance per distribution object and
// pays for it per particle anyway -- Sigma.determinant() and Sigma.inverse() inside every pdf()
// call (src/statistics.cc.cpp:176-177, 301, 306) -- so a model whose covariance differs from
// particle to particle (Rao-Blackwellised / auxiliary filters) costs it nothing extra.  Here the
// shared-covariance kernels factor once on the host; these two serve the per-particle case:
//     cholesky_batched_kernel   Sigma_i = L_i L_i^T, logdet_i = 2 sum ln L_i,jj, info_i
//     logpdf_percov_kernel      out_i = log p(x_i; mu_i, Sigma_i), MVN or Student-t
//
// One lane = one matrix, d a template parameter so that every loop unrolls.  The lower triangle
// lives in registers up to d = 8 (36 doubles) and above that in LDS as [element][lane (+1 pad)]
// (consecutive lanes on consecutive banks: conflict-free; d = 16 needs 136 doubles per lane,
// which no register file holds); it is factored in place column by column, and for the density
// the forward substitution z = L^-1 (x - mu) follows in registers.  Every sum is an fma chain in
// index order and the only other roundings are sqrt and divide, so oracle/cusmc_oracle.c gets
// the same L bit for bit.
//
// Algorithmic HBM bytes per matrix: 8 d^2 in, 8 d^2 (+ 12) out for the factorisation; 8 d^2 +
// 8 d (+ 8 d) + 8 for the density.  d^3/3 LDS-operand FMAs per lane: LDS-issue-bound, not
// HBM-bound, above d ~ 8.
#include "smallops.h"

namespace cusmc {

namespace {

constexpr int kPercovLanes = 64;   // one wave per workgroup
constexpr int kSlabStride = 65;    // an LDS slab is [element][64 lanes + 1]: a lane's own accesses are conflict-free
                                   // (consecutive lanes, consecutive words) and so, nearly, are the transposed ones
                                   // of the cooperative store below

constexpr int tri_index(int r, int c) { return r * (r + 1) / 2 + c; }
template <int D>
constexpr bool percov_in_regs() { return D <= 8; }  // 36 doubles at d = 8; LDS above

// The lane's lower triangle: registers (every index below is a compile-time constant once the
// loops are unrolled) or the lane's column of the slab.
template <int D, bool REG>
struct Triangle {
  double reg[REG ? D * (D + 1) / 2 : 1];
  double *lds;
  __device__ __forceinline__ double get(int e) const { return REG ? reg[e] : lds[e * kSlabStride]; }
  __device__ __forceinline__ void set(int e, double v)
  {
    if (REG) reg[e] = v; else lds[e * kSlabStride] = v;
  }
};

// Loads the lower triangle of S (row-major D x D) and factors it in place, column by column: the
// D - j - 1 elements below a pivot are independent of one another, which is the instruction-level
// parallelism a lane has (each element's own sum is an fma chain in k order, as in the oracle).
// Returns 0 or 1 + the index of the first pivot that is not positive (the factor is then not
// finite from that row on); *logdet = 2 sum ln L_jj.
template <int D, bool REG>
__device__ __forceinline__ int cholesky_lane(const double *__restrict__ S, Triangle<D, REG> &a, double *logdet)
{
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) a.set(tri_index(r, c), S[r * D + c]);
  int bad = 0;
  double ld = 0.0;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double s = a.get(tri_index(j, j));
#pragma unroll
    for (int k = 0; k < j; ++k) {
      const double v = a.get(tri_index(j, k));
      s = fma(-v, v, s);
    }
    if (!(s > 0.0) && bad == 0) bad = j + 1;
    const double ljj = sqrt(s);
    a.set(tri_index(j, j), ljj);
    ld += ln_pos(ljj);
#pragma unroll
    for (int r = j + 1; r < D; ++r) {
      double t = a.get(tri_index(r, j));
#pragma unroll
      for (int k = 0; k < j; ++k) t = fma(-a.get(tri_index(r, k)), a.get(tri_index(j, k)), t);
      a.set(tri_index(r, j), t / ljj);
    }
  }
  *logdet = bad ? __builtin_nan("") : 2.0 * ld;
  return bad;
}

}  // namespace

// The factors leave through LDS: a wave's 64 matrices are contiguous in L (64 D^2 doubles), so instead
// of each lane storing its own rows 8 bytes at a time (64 distinct lines per instruction) the wave
// stores the block as D^2 fully coalesced 512-byte instructions.  Register variant: staged as
// [lane][D^2 + 1]; LDS variant: read back transposed from the slab itself.
template <int D>
__global__ __launch_bounds__(kPercovLanes) void cholesky_batched_kernel(
    const double *__restrict__ S, long N, double *__restrict__ L, double *__restrict__ logdet,
    int *__restrict__ info)
{
  constexpr bool REG = percov_in_regs<D>();
  constexpr int DD = D * D;
  extern __shared__ double slab[];
  Triangle<D, REG> a;
  a.lds = slab + threadIdx.x;
  const int lane = threadIdx.x;
  for (long i0 = (long)blockIdx.x * kPercovLanes; i0 < N; i0 += (long)gridDim.x * kPercovLanes) {
    const long i = i0 + lane;
    double ld = 0.0;
    int bad = 0;
    if (i < N) bad = cholesky_lane<D, REG>(S + i * DD, a, &ld);
    if constexpr (REG) {
      double *mine = slab + lane * (DD + 1);
#pragma unroll
      for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) mine[r * D + c] = c <= r ? a.get(tri_index(r, c)) : 0.0;
    }
    __syncthreads();  // (one wave: orders the LDS writes before the transposed reads)
    const long total = (N - i0 < kPercovLanes ? N - i0 : (long)kPercovLanes) * DD;
    double *dst = L + i0 * DD;
#pragma unroll 4
    for (int c = 0; c < DD; ++c) {
      const int flat = c * kPercovLanes + lane;
      const int m = flat / DD, e = flat - m * DD;
      double v;
      if constexpr (REG) {
        v = slab[m * (DD + 1) + e];
      } else {
        const int r = e / D, col = e - r * D;
        v = col <= r ? slab[tri_index(r, col) * kSlabStride + m] : 0.0;
      }
      if (flat < total) dst[flat] = v;
    }
    __syncthreads();  // the slab is rewritten by the next batch
    if (i < N) {
      if (logdet) logdet[i] = ld;
      if (info) info[i] = bad;
    }
  }
}

// mu: N x ldmu (ldmu >= d), or one shared vector when ldmu == 0, or NULL (zero mean).
// ep.lognorm is the normalising constant WITHOUT the determinant term; -logdet_i / 2 is added here.
template <int D>
__global__ __launch_bounds__(kPercovLanes) void logpdf_percov_kernel(
    const double *__restrict__ X, long N, long ldx, const double *__restrict__ mu, long ldmu,
    const double *__restrict__ S, Epilogue ep, double *__restrict__ out, int *__restrict__ info)
{
  constexpr bool REG = percov_in_regs<D>();
  extern __shared__ double slab[];
  Triangle<D, REG> a;
  a.lds = slab + threadIdx.x;
  for (long i = (long)blockIdx.x * kPercovLanes + threadIdx.x; i < N; i += (long)gridDim.x * kPercovLanes) {
    double ld;
    const int bad = cholesky_lane<D, REG>(S + i * (D * D), a, &ld);
    const double *x = X + i * ldx;
    const double *m = mu ? mu + i * ldmu : nullptr;
    double z[D];
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {  // forward substitution, z_j = (r_j - sum_k<j L_jk z_k) / L_jj
      double t = m ? x[j] - m[j] : x[j];
#pragma unroll
      for (int k = 0; k < j; ++k) t = fma(-a.get(tri_index(j, k)), z[k], t);
      z[j] = t / a.get(tri_index(j, j));
      q = fma(z[j], z[j], q);
    }
    double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p_nonneg(q * ep.inv_nu) : ep.lognorm - 0.5 * q;
    lp -= 0.5 * ld;
    out[i] = bad ? __builtin_nan("") : (ep.out_density ? exp(lp) : lp);
    if (info) info[i] = bad;
  }
}

bool percov_supported(int d) { return d >= 1 && d <= 16; }

static long percov_blocks(int64_t N, int num_cus)
{
  long blocks = (N + kPercovLanes - 1) / kPercovLanes;
  const long cap = (long)num_cus * 16;
  return blocks > cap ? cap : blocks;
}

template <int D>
static size_t percov_lds_bytes(bool factor_out)
{
  if (percov_in_regs<D>()) return factor_out ? (size_t)kPercovLanes * (D * D + 1) * sizeof(double) : 0;  // the store's staging buffer
  return (size_t)(D * (D + 1) / 2) * kSlabStride * sizeof(double);
}

template <int D>
static hipError_t launch_chol(const double *S, int64_t N, double *L, double *logdet, int *info, int num_cus,
                              hipStream_t stream)
{
  const size_t lds = percov_lds_bytes<D>(true);
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(cholesky_batched_kernel<D>), lds, lds_configured); e != hipSuccess) return e;
  hipLaunchKernelGGL(cholesky_batched_kernel<D>, dim3((unsigned)percov_blocks(N, num_cus)), dim3(kPercovLanes), lds,
                     stream, S, (long)N, L, logdet, info);
  return hipGetLastError();
}

template <int D>
static hipError_t launch_lp(const double *X, int64_t N, int64_t ldx, const double *mu, int64_t ldmu, const double *S,
                            const Epilogue &ep, double *out, int *info, int num_cus, hipStream_t stream)
{
  const size_t lds = percov_lds_bytes<D>(false);
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(logpdf_percov_kernel<D>), lds, lds_configured); e != hipSuccess) return e;
  hipLaunchKernelGGL(logpdf_percov_kernel<D>, dim3((unsigned)percov_blocks(N, num_cus)), dim3(kPercovLanes), lds,
                     stream, X, (long)N, (long)ldx, mu, (long)ldmu, S, ep, out, info);
  return hipGetLastError();
}

#define CUSMC_PERCOV_DIMS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

hipError_t launch_cholesky_batched(const double *S, int64_t N, int d, double *L, double *logdet, int *info,
                                   int num_cus, hipStream_t stream)
{
  if (N <= 0) return hipSuccess;
#define CUSMC_CASE(D) case D: return launch_chol<D>(S, N, L, logdet, info, num_cus, stream);
  switch (d) { CUSMC_PERCOV_DIMS(CUSMC_CASE) }
#undef CUSMC_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_logpdf_percov(const double *X, int64_t N, int64_t ldx, const double *mu, int64_t ldmu,
                                const double *S, int d, const Epilogue &ep, double *out, int *info, int num_cus,
                                hipStream_t stream)
{
  if (N <= 0) return hipSuccess;
#define CUSMC_CASE(D) case D: return launch_lp<D>(X, N, ldx, mu, ldmu, S, ep, out, info, num_cus, stream);
  switch (d) { CUSMC_PERCOV_DIMS(CUSMC_CASE) }
#undef CUSMC_CASE
  return hipErrorInvalidValue;
}

}  // namespace cusmc
