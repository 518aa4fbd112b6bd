// Batched multivariate Normal / Student-t log-density, any d (<= 639), fp64 -- the shape-agnostic
// path (d not a multiple of 16, unaligned or odd-stride batches, tiny d).  Same contract as
// kernels/logpdf_mfma.hip:   z = bias + M (x - shift),  q = z.z,  out = epilogue(q);
// same reference functions replaced (src/statistics.cc.cpp:171-196, :295-324;
// src/mvn_dist.cu.cpp:455-668; src/mvt_dist.cu.cpp:356-571).
//
// Mapping: lane = particle.  A workgroup stages a tile of T particles through LDS with
// fully coalesced 8-byte loads (consecutive lanes read consecutive doubles of the flat
// batch), subtracting `shift` on the way in; each lane then walks its own row.  Rows are padded
// to an odd number of doubles, so a wave's ds_read_b64 of column k touches 32 distinct banks
// per half-wave.  M is read with wave-uniform addresses, i.e. scalar loads through the constant
// cache: the factor costs no VALU or LDS bandwidth here.  For the triangular (centred) form the
// inner loop stops at the diagonal.
//
// Small d is HBM-bound (8d+8 bytes against ~d^2/2 FMAs per particle); this kernel is the
// fallback for large odd d, not the tuned path.
#include <hip/hip_runtime.h>

#include "smallops.h"

namespace cusmc {

template <int T>
__global__ __launch_bounds__(T) void logpdf_generic_kernel(
    const double *__restrict__ X, long N, long ldx, int d, int tri, const double *__restrict__ M,
    const double *__restrict__ shift, const double *__restrict__ bias, Epilogue ep,
    double *__restrict__ out, long num_tiles)
{
  extern __shared__ double sR[];  // T rows x stride
  const int stride = d | 1;
  const bool flat = (ldx == d);

  for (long tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const long base = tile * T;
    const long rows = (N - base) < T ? (N - base) : T;
    const long elems = rows * d;
    __syncthreads();  // the previous tile's readers are done with sR
    if (flat) {
      const double *src = X + base * ldx;
      int row = (int)(threadIdx.x / d), col = (int)(threadIdx.x % d);
      const int drow = T / d, dcol = T % d;
      for (long e = threadIdx.x; e < elems; e += T) {
        sR[row * stride + col] = src[e] - shift[col];
        row += drow;
        col += dcol;
        if (col >= d) { col -= d; ++row; }
      }
    } else {
      int row = (int)(threadIdx.x / d), col = (int)(threadIdx.x % d);
      const int drow = T / d, dcol = T % d;
      for (long e = threadIdx.x; e < elems; e += T) {
        sR[row * stride + col] = X[(base + row) * ldx + col] - shift[col];
        row += drow;
        col += dcol;
        if (col >= d) { col -= d; ++row; }
      }
    }
    __syncthreads();
    if ((long)threadIdx.x < rows) {
      const double *r = sR + threadIdx.x * stride;
      double q = 0.0;
      for (int j = 0; j < d; ++j) {
        const double *Mj = M + (long)j * d;
        const int kend = tri ? j + 1 : d;
        double z0 = bias[j], z1 = 0.0;
        int k = 0;
        for (; k + 1 < kend; k += 2) {
          z0 = fma(Mj[k], r[k], z0);
          z1 = fma(Mj[k + 1], r[k + 1], z1);
        }
        if (k < kend) z0 = fma(Mj[k], r[k], z0);
        const double z = z0 + z1;
        q = fma(z, z, q);
      }
      out[base + threadIdx.x] = finish_generic(q, ep);
    }
  }
}

// d < 16: one lane per particle, the row in REGISTERS (no LDS staging): D doubles per lane through
// 16-byte loads where the row allows them, M / shift / bias through wave-uniform (scalar) loads.
// Same operation order as the staged kernel above and as pf_step.hip, so the three agree bitwise.
template <int D>
__global__ __launch_bounds__(256) void logpdf_small_kernel(
    const double *__restrict__ X, long N, long ldx, int tri, const double *__restrict__ M,
    const double *__restrict__ shift, const double *__restrict__ bias, Epilogue ep,
    double *__restrict__ out)
{
  const long stride = (long)gridDim.x * 256;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < N; i += stride) {
    const double *row = X + i * ldx;
    double r[D];
    if ((D % 2 == 0) && (ldx % 2 == 0) && ((uintptr_t)X % 16 == 0)) {  // uniform
#pragma unroll
      for (int k = 0; k < D; k += 2) {
        typedef double v2 __attribute__((ext_vector_type(2)));
        const v2 t = *reinterpret_cast<const v2 *>(row + k);
        r[k] = t[0];
        if (k + 1 < D) r[k + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) r[k] = row[k];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) r[k] = r[k] - shift[k];
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
    out[i] = finish_generic(q, ep);
  }
}

template <int D>
static hipError_t launch_small(const double *X, int64_t N, int64_t ldx, bool tri, const double *M,
                               const double *shift, const double *bias, const Epilogue &ep, double *out,
                               int num_cus, hipStream_t stream)
{
  long blocks = (N + 255) / 256;
  const long cap = (long)num_cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(logpdf_small_kernel<D>, dim3((unsigned)blocks), dim3(256), 0, stream, X, (long)N, (long)ldx,
                     (int)tri, M, shift, bias, ep, out);
  return hipGetLastError();
}

bool generic_supported(int d) { return d >= 1 && (size_t)32 * (d | 1) * 8 <= 160 * 1024; }  // (d <= 639; CUSMC_MAX_DIM is below that)

template <int T>
static hipError_t launch_t(const double *X, int64_t N, int64_t ldx, int d, bool tri,
                           const double *M, const double *shift, const double *bias,
                           const Epilogue &ep, double *out, int num_cus, hipStream_t stream)
{
  const size_t lds_bytes = (size_t)T * (d | 1) * sizeof(double);
  auto kern = logpdf_generic_kernel<T>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  const long num_tiles = (N + T - 1) / T;
  int per_cu = (int)((160 * 1024) / lds_bytes);
  const int max_per_cu = 2048 / T > 8 ? 8 : 2048 / T;
  per_cu = per_cu > max_per_cu ? max_per_cu : (per_cu < 1 ? 1 : per_cu);
  long blocks = (long)num_cus * per_cu;
  if (blocks > num_tiles) blocks = num_tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(T), lds_bytes, stream, X, (long)N,
                     (long)ldx, d, (int)tri, M, shift, bias, ep, out, num_tiles);
  return hipGetLastError();
}

hipError_t launch_logpdf_generic(const double *X, int64_t N, int64_t ldx, int d, bool tri,
                                 const double *M, const double *shift, const double *bias,
                                 const Epilogue &ep, double *out, int num_cus, hipStream_t stream)
{
  if (N <= 0) return hipSuccess;
  if (!generic_supported(d)) return hipErrorInvalidValue;
#define CUSMC_SMALL(D) case D: return launch_small<D>(X, N, ldx, tri, M, shift, bias, ep, out, num_cus, stream);
  switch (d) {
    CUSMC_SMALL(1) CUSMC_SMALL(2) CUSMC_SMALL(3) CUSMC_SMALL(4) CUSMC_SMALL(5) CUSMC_SMALL(6) CUSMC_SMALL(7)
    CUSMC_SMALL(8) CUSMC_SMALL(9) CUSMC_SMALL(10) CUSMC_SMALL(11) CUSMC_SMALL(12) CUSMC_SMALL(13)
    CUSMC_SMALL(14) CUSMC_SMALL(15)
    default: break;
  }
#undef CUSMC_SMALL
  const size_t row_bytes = (size_t)(d | 1) * 8;
  if (256 * row_bytes <= 64 * 1024)
    return launch_t<256>(X, N, ldx, d, tri, M, shift, bias, ep, out, num_cus, stream);
  if (128 * row_bytes <= 64 * 1024)
    return launch_t<128>(X, N, ldx, d, tri, M, shift, bias, ep, out, num_cus, stream);
  if (64 * row_bytes <= 160 * 1024)
    return launch_t<64>(X, N, ldx, d, tri, M, shift, bias, ep, out, num_cus, stream);
  return launch_t<32>(X, N, ldx, d, tri, M, shift, bias, ep, out, num_cus, stream);  // (d >= 320: half a wave per tile)
}

}  // namespace cusmc
