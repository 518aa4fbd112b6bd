// Batched multivariate Normal / Student-t log-density for d = 16*NB on gfx950, fp64.
//
// Replaces the reference's three-launch pdf pipeline (mvn_pdf_kernel_y_minus_Fmu ->
// mvn_pdf_kernel_Einv_alpha -> mvn_pdf_kernel, src/mvn_dist.cu.cpp:455-668, and the mvt twins,
// src/mvt_dist.cu.cpp:356-571) and, on the CPU side, the per-particle
// MultiVariateNormalDistribution::pdf / MultiVariateTStudentDistribution::pdf
// (src/statistics.cc.cpp:171-196, :295-324) called from reweight_G (src/mcmc.cpp:193-215).
//
// Formulation.  Sigma = L L^T is factored ONCE on the host; with W = L^-1 the Mahalanobis
// form is q = |W (x - m)|^2.  Over a batch that is Z = R W^T, a [N x d] x [d x d] GEMM whose
// right factor is lower triangular -- so it runs on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64), skipping the all-zero 16x16 blocks above the diagonal.  Z is never
// written: each wave squares and row-sums its accumulators and emits 8 bytes per particle.
//
// Mapping (one wave = one tile of 16 particles at a time, persistent over tiles):
//   A operand  lane (p = lane&15, h = lane>>4) holds r[p][k], k = 16*kb + pi(s,h), for k-step s
//              of k-block kb, where pi(s,h) = 2h + (s&1) + 8(s>>1).  The k order inside a block
//              is free (it is a summation index) and this one lets each lane fetch its four
//              values of a block with two 16-byte loads, 64 contiguous bytes per particle per
//              load instruction: X goes HBM -> VGPR once, coalesced, with no LDS round trip.
//   B operand  lane (j = lane&15, h) holds M[16*cb + j][16*kb + pi(s,h)]: the factor, packed on
//              the host in exactly this order (mfma_pack_frags) and staged ONCE per workgroup in
//              LDS (20 KB for the triangular d = 64 factor); one conflict-free ds_read_b64 per
//              MFMA.
//   C/D        lane (j, g = lane>>4), register r  ->  particle g + 4r, output dim 16*cb + j.
//   Epilogue   per lane sum_cb acc[cb][r]^2, then a 16-lane DPP reduction over j.
//
// Roofline: 8d + 8 algorithmic bytes per particle; at d = 64 the kernel needs 40
// MFMAs of 2048 flop per 16 particles.
#include <hip/hip_runtime.h>

#include "../launch.h"
#include "../../../include/cusmc_hip.h"

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int pi_k(int s, int h) { return 2 * h + (s & 1) + 8 * (s >> 1); }

int mfma_num_frags(int nb, bool tri) { return tri ? 4 * nb * (nb + 1) / 2 : 4 * nb * nb; }

bool mfma_supported(int d, const void *X, int64_t ldx)
{
  if (d % 16 != 0) return false;
  const int nb = d / 16;
  if (!(nb == 1 || nb == 2 || nb == 3 || nb == 4 || nb == 6 || nb == 8)) return false;
  // the A-operand loads are 16-byte vector loads
  return ((uintptr_t)X % 16 == 0) && (ldx % 2 == 0);
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

// Sum over the 16 lanes of a DPP row.  v_add_f64 has no DPP form on gfx9, so each step moves
// the two 32-bit halves with v_mov_b32_dpp.  Controls: quad_perm [1,0,3,2] = 0xB1,
// quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v)
{
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}

__device__ __forceinline__ double finish(double q, const Epilogue &ep)
{
  double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p(q * ep.inv_nu)
                                     : ep.lognorm - 0.5 * q;
  return ep.out_density ? exp(lp) : lp;
}

template <int NB, bool TRI>
__global__ __launch_bounds__(256) void logpdf_mfma_kernel(
    const double *__restrict__ X, long N, long ldx, const double *__restrict__ frags,
    const double *__restrict__ shift, const double *__restrict__ bias, Epilogue ep,
    double *__restrict__ out, long num_tiles)
{
  constexpr int NFRAG = TRI ? 4 * NB * (NB + 1) / 2 : 4 * NB * NB;
  extern __shared__ double lds[];
  double *sF = lds;                 // NFRAG x 64
  double *sShift = lds + NFRAG * 64;  // 16*NB
  double *sBias = sShift + 16 * NB;   // 16*NB

  // Stage the factor: all of a chunk's 16-byte loads are issued before the first LDS write, so
  // the prologue costs one memory round trip per chunk, not one per element.
  {
    constexpr int NV = NFRAG * 32;  // 16-byte elements
    constexpr int CH = 8;
    const v2d *g = reinterpret_cast<const v2d *>(frags);
    v2d *l = reinterpret_cast<v2d *>(sF);
    for (int base = 0; base < NV; base += CH * 256) {
      v2d tmp[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = base + c * 256 + (int)threadIdx.x;
        if (i < NV) tmp[c] = g[i];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = base + c * 256 + (int)threadIdx.x;
        if (i < NV) l[i] = tmp[c];
      }
    }
    if (threadIdx.x < 16 * NB) {
      sShift[threadIdx.x] = shift[threadIdx.x];
      sBias[threadIdx.x] = bias[threadIdx.x];
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int p = lane & 15, h = lane >> 4;
  const long nwaves = (long)gridDim.x * 4;
  long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= num_tiles) return;

  auto load_tile = [&](long t, v2d(&a)[NB][2]) {
    long row = t * 16 + p;
    row = row < N ? row : N - 1;  // tail rows re-read the last particle; their stores are masked
    const double *src = X + row * ldx + 2 * h;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      a[kb][0] = *reinterpret_cast<const v2d *>(src + 16 * kb);
      a[kb][1] = *reinterpret_cast<const v2d *>(src + 16 * kb + 8);
    }
  };

  // The factor fragments are loop-invariant LDS reads; left alone, hipcc hoists all of them out
  // of the tile loop into (spilled) registers and occupancy drops to one wave per SIMD.  An
  // opaque per-tile lane offset keeps them as in-loop ds_read_b64.
  int lds_lane = lane;

  auto compute_tile = [&](long t, const v2d(&a_in)[NB][2]) {
    asm volatile("" : "+v"(lds_lane));
    v4d acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const double b = sBias[16 * cb + p];
      acc[cb] = v4d{b, b, b, b};
    }
    int f = 0;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = a_in[kb][s >> 1][s & 1] - sShift[16 * kb + pi_k(s, h)];
#pragma unroll
        for (int cb = TRI ? kb : 0; cb < NB; ++cb, ++f)
          acc[cb] =
              __builtin_amdgcn_mfma_f64_16x16x4f64(a, sF[f * 64 + lds_lane], acc[cb], 0, 0, 0);
      }
    }
    double q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) s = fma(acc[cb][r], acc[cb][r], s);
      q[r] = row16_sum(s);
    }
    // every lane of a row now holds the four totals of particles g + 4r; lanes j < 4 finish
    // and store particle g + 4j.
    if (p < 4) {
      const double qq = p == 0 ? q[0] : p == 1 ? q[1] : p == 2 ? q[2] : q[3];
      const long row = t * 16 + h + 4 * p;
      if (row < N) out[row] = finish(qq, ep);
    }
  };

  // Two register sets, roles swapped each half-iteration: the loads of the next tile are in
  // flight while the current one runs on the matrix cores, with no register copies.  The
  // prefetch is unconditional -- past the end it re-reads the current tile (an L2 hit, result
  // unused) -- because a branch around it makes hipcc's s_waitcnt placement assume the
  // no-prefetch path and wait for the prefetched loads at the head of every tile.
  v2d a0[NB][2], a1[NB][2];
  load_tile(tile, a0);
  while (true) {
    const long t1 = tile + nwaves;
    load_tile(t1 < num_tiles ? t1 : tile, a1);
    compute_tile(tile, a0);
    if (t1 >= num_tiles) break;
    const long t2 = t1 + nwaves;
    load_tile(t2 < num_tiles ? t2 : t1, a0);
    compute_tile(t1, a1);
    if (t2 >= num_tiles) break;
    tile = t2;
  }
}

template <int NB, bool TRI>
static hipError_t launch_nb(const double *X, int64_t N, int64_t ldx, const double *frags,
                            const double *shift, const double *bias, const Epilogue &ep,
                            double *out, int num_cus, hipStream_t stream)
{
  constexpr int NFRAG = TRI ? 4 * NB * (NB + 1) / 2 : 4 * NB * NB;
  const size_t lds_bytes = (size_t)(NFRAG * 64 + 32 * NB) * sizeof(double);
  const long num_tiles = (N + 15) / 16;
  auto kern = logpdf_mfma_kernel<NB, TRI>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  // persistent grid: as many 4-wave workgroups per CU as LDS admits (<= 3), capped by the work
  int per_cu = (int)((160 * 1024) / lds_bytes);
  per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
  long blocks = (long)num_cus * per_cu;
  const long need = (num_tiles + 3) / 4;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, X, (long)N,
                     (long)ldx, frags, shift, bias, ep, out, num_tiles);
  return hipGetLastError();
}

hipError_t launch_logpdf_mfma(const double *X, int64_t N, int64_t ldx, int d, bool tri,
                              const double *frags, const double *shift, const double *bias,
                              const Epilogue &ep, double *out, int num_cus, hipStream_t stream)
{
  if (N <= 0) return hipSuccess;
#define CUSMC_CASE(nb)                                                                          \
  case nb:                                                                                      \
    return tri ? launch_nb<nb, true>(X, N, ldx, frags, shift, bias, ep, out, num_cus, stream)   \
               : launch_nb<nb, false>(X, N, ldx, frags, shift, bias, ep, out, num_cus, stream);
  switch (d / 16) {
    CUSMC_CASE(1)
    CUSMC_CASE(2)
    CUSMC_CASE(3)
    CUSMC_CASE(4)
    CUSMC_CASE(6)
    CUSMC_CASE(8)
  }
#undef CUSMC_CASE
  return hipErrorInvalidValue;
}

}  // namespace cusmc
