// Proposal draws for 128 < d <= 256 (NB = ceil(d/16) = 9 .. 16 blocks) on the f64 matrix cores:
//     X_out^T = Q Xi^T [.* C] + G X_gathered^T          (propagate_K, src/mcmc.cpp:90-160)
//     X_out^T = Q Xi^T [.* C] + m0                       (initialize / R-level draws, src/mcmc.cpp:44-88)
// Replaces, for these d, the one-workgroup-per-particle propagate_row_kernel of round 1 (every particle
// re-read both d x d factors from L2: 1 MB per particle at d = 256) and, on the reference's side, the
// one-block-per-particle-per-tile GEMV pair mvn_sample_Gmu_kernel / mvn_sample_kernel
// (src/mvn_dist.cu.cpp:33-172; mvt twins src/mvt_dist.cu.cpp:84-223).
//
// The factors (512 KB each at d = 256) fit neither LDS nor registers, so -- as in logpdf_mfma_wide.hip --
// the OUTPUT dimension is split over the eight waves of a workgroup and the B operands of a group of 64
// particles (four 16-particle tiles) are staged once in LDS for all of them:
//   fill      slab (kb, h2, t) holds, for lane (p, h), the operands of k-steps 2 h2 and 2 h2 + 1 of k-block kb for
//             particle 16 t + p: columns 16 kb + 8 h2 + 2 h and + 1 -- exactly ONE Box-Muller pair (Philox block
//             sub = 8 kb + 4 h2 + h: the RNG contract keys a block by component pair), resp. one 16-byte piece of
//             the ancestor's row.  Wave w fills tile w % 4, half w / 4 of every k-block; a lane writes its 16
//             bytes at slab + 16 lane; a compute wave reads a slab back with one conflict-free ds_read_b128.
//   multiply  wave w owns output blocks w and NB - 1 - w (one block where that is not above w) for the four tiles; the A fragments of its blocks stream
//             from L2 (mfma_pack_frags order, 512 contiguous bytes per fragment, buffer loads) one k-block ahead
//             of their use, each feeding FOUR MFMAs: 2 blocks x 4 tiles x 4 NB MFMAs per factor, group and wave.
//   the two products take TURNS on one set of slabs (128 KB at d = 256): normals -> Q Xi -> [Student-t: scale] ->
//             gathered rows -> += G X.  Round 2's first version kept both operand sets in LDS at once, which
//             capped a group at 32 particles: every fragment fed two MFMAs and the eight waves pulled 1 MB from
//             L2 per group -- 16 GB per 5e5 x 256 launch, ~10 TB/s, which is what the launch took (ablation: 1.83
//             ms with the MFMAs removed, 2.50 ms with the fill removed, against a 1.68 ms MFMA floor).  Four tiles
//             halve that traffic.
//   epilogue  lane (p, h), register r holds output dim 16 cb + h + 4 r of particle p: Student-t scaling by
//             sqrt(nu / chi2) per component (src/statistics.cc.cpp:385-386, 411; chi_square_batch, smallops.h),
//             + g .* x (diagonal G) or + m0, stored 8 bytes per lane.
// Three barriers per group with a dense G (four for Student-t), two without.  The fill (RNG: VALU) and the multiply
// (MFMA) of one workgroup do not overlap; an f64 MFMA blocks VALU issue on its SIMD anyway.
// Algorithmic bytes per particle 16 d + 4; flops 2 x 2 x (16 NB)^2 (dense G) -- MFMA-bound: 1.7 ms for 5e5 x 256
// at the 77.7 TF peak.
//
// d that is not a multiple of 16 runs with the factors zero-padded to 16 NB on the host (PAD): normals
// for pairs past d are not drawn, gathered columns past d are zeroed, outputs past d are not stored.
//
// TRIQ (propagate_mfma_wide_tri.hip: this file with CUSMC_TRIQ = 1): Q is LOWER TRIANGULAR (a Cholesky factor), its
// fragments packed triangular (mfma_pack_frags, tri = true).  Output block cb of Q Xi needs the k-blocks kb <= cb only;
// the pair (w, NB - 1 - w) costs NB + 1 block-products on every wave -- 136 per tile at d = 256 instead of 256 -- in
// two phases: both blocks up to kb = w, then the upper one alone.  G x stays dense.
#include <type_traits>

#include "smallops.h"

#ifndef CUSMC_TRIQ
#define CUSMC_TRIQ 0
#endif

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));  // rows are 8-byte aligned in general

constexpr int kWideTiles = 4;          // 16-particle tiles per group
constexpr int kWideGroup = 16 * kWideTiles;

#if !CUSMC_TRIQ
bool propagate_mfma_wide_supported(int d, const void *X_prev, const void *X_out)
{
  return d > 128 && d <= 256 && (uintptr_t)X_prev % 8 == 0 && (uintptr_t)X_out % 8 == 0;
}
#endif

// one set of slabs (the two products take turns); the Student-t kernel also parks 8 waves x 16 KB of accumulators there
static size_t propagate_wide_lds_bytes(int nb, bool mvt)
{
  const size_t slabs = (size_t)nb * 2 * kWideTiles * 1024, park = 8 * 16384;
  return (mvt && park > slabs ? park : slabs) + (mvt ? 8 * sizeof(ChiQueue) : 0);  // (+ a queue per wave for its open chi^2 draws)
}

// MODE 0: x = [diag(c)] Q xi + m0     (tail = m0, d doubles)
//      1: x = [diag(c)] Q xi + G x_prev[a]   (tail = fragments of G)
//      4: x = [diag(c)] Q xi + g .* x_prev[a]   (tail = diag(G), d doubles)
// Calibration hook (scripts/calib/pw_phases.hip; off in the library): wave 0 of every workgroup adds the s_memtime
// ticks between the phase boundaries of a group to g_pw_phases[8 blockIdx + phase].
#ifdef CUSMC_PW_PHASES
__device__ unsigned long long g_pw_phases[8 * 1024];
#define PW_STAMP(k)                                                                      \
  do {                                                                                   \
    const unsigned long long now_ = __builtin_readcyclecounter();                        \
    if (threadIdx.x == 0 && (k) > 0) g_pw_phases[8 * blockIdx.x + (k)] += now_ - pw_t_;  \
    pw_t_ = now_;                                                                        \
  } while (0)
#else
#define PW_STAMP(k) do { } while (0)
#endif

template <int NB, bool MVT, int MODE, bool PAD, bool TRIQ>
__global__ __launch_bounds__(512) void propagate_wide_kernel(
    float nu, const double *__restrict__ X_prev, const uint32_t *__restrict__ a,
    const double *__restrict__ fragsQ, const double *__restrict__ tail, int d, double scale, uint32_t k0,
    uint32_t k1, uint32_t step, uint32_t domain, uint32_t first, uint32_t count, double *__restrict__ X_out,
    long num_groups)
{
  constexpr int T = kWideTiles;
  static_assert(T == 4, "the fill below gives wave w tile w % 4, half (w / 4) & 1 and every k-block");
  constexpr bool HAS_G = MODE == 1;
  extern __shared__ double lds[];
  double *sB = lds;  // NB x 2 x T slabs of 128 doubles: the normals, then (HAS_G) the gathered rows

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Student-t: this wave's queue for its open chi^2 draws (smallops.h: ChiQueue), behind the slabs / the parking area
  constexpr size_t kSlabDoubles = (size_t)NB * 2 * T * 128 > 8 * 2048 ? (size_t)NB * 2 * T * 128 : 8 * 2048;
  ChiQueue *const chi_q = MVT ? reinterpret_cast<ChiQueue *>(lds + kSlabDoubles) + w : nullptr;
  if (MVT && lane == 0) chi_q->count = 0;  // (the first use lies behind several __syncthreads())
  const int p = lane & 15, h = lane >> 4;
  const ChiSquare cs = chi_setup(MVT ? nu : 2.0f);
  // this wave's output blocks: w, and NB - 1 - w for the NB - 8 waves that carry two (with a triangular Q a pair costs
  // (w + 1) + (NB - w) k-blocks whatever w is)
  const int cb0 = w, cb1 = NB - 1 - w;
  const bool two = w < NB - 8;
  // this wave's share of a fill: slabs w + 8 kb, kb = 0 .. NB - 1 -- tile ft, half fh2 of every k-block
  const int ft = w & 3, fh2 = w >> 2;

#ifdef CUSMC_PW_PHASES
  unsigned long long pw_t_ = 0;
#endif
  // the ancestor index of the lane's fill particle, fetched one group AHEAD (Normal, dense G): the gather's addresses
  // depend on it, vmcnt retires in order, and a load issued at the top of a group would wait for the 32 stores of the
  // previous group's epilogue to drain first
  auto ancestor_of = [&](long gg) -> uint32_t {
    const long fl = gg * kWideGroup + 16 * ft + p;
    const long cl = fl < (long)count ? fl : (long)count - 1;
    return a ? a[cl] : first + (uint32_t)cl;
  };
  uint32_t anc_ahead = 0;
  if constexpr (HAS_G && !MVT) {
    if ((long)blockIdx.x < num_groups) anc_ahead = ancestor_of(blockIdx.x);
  }
  for (long g = blockIdx.x; g < num_groups; g += gridDim.x) {
    PW_STAMP(0);
    const long base = g * kWideGroup;  // first local row of the group
    const long flocal = base + 16 * ft + p;
    const bool flive = flocal < (long)count;
    const uint32_t fgi = first + (uint32_t)(flive ? flocal : (long)count - 1);
    // ---- the ancestor's row.  Normal: requested first, its pieces travel while the normals are drawn and wait in
    // 2 NB registers until the slabs are free.  Student-t: requested when the slabs are free (the chi-square
    // constants on top of those registers would spill in the multiply; the exposed round trip is ~1.5 %) ----------
    v2d xg[HAS_G ? NB : 1];
    auto gather = [&]() {
      const uint32_t anc = MVT ? (a ? a[flive ? flocal : (long)count - 1] : fgi) : anc_ahead;
      const double *row = X_prev + (long)anc * d;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const int col = 16 * kb + 8 * fh2 + 2 * h;
        if (!PAD || col + 1 < d) {
          xg[kb] = *reinterpret_cast<const v2d_a8 *>(row + col);
        } else {
          xg[kb] = v2d{col < d ? row[col] : 0.0, 0.0};
        }
      }
    };
    if constexpr (HAS_G && !MVT) gather();
    // ---- fill: the normals ----------------------------------------------------------------------------------
#pragma unroll 1
    for (int kb = 0; kb < NB; ++kb) {
      const int col = 16 * kb + 8 * fh2 + 2 * h;  // the pair's first column
      double z0 = 0.0, z1 = 0.0;
      if (!PAD || col < d) {
        normal_pair(philox4x32_10(fgi, (uint32_t)(8 * kb + 4 * fh2 + h), step, domain, k0, k1), z0, z1);
        z0 *= scale;
        z1 = (!PAD || col + 1 < d) ? z1 * scale : 0.0;
      }
      reinterpret_cast<v2d *>(sB + (w + 8 * kb) * 128)[lane] = v2d{z0, z1};
    }
    PW_STAMP(1);
    __syncthreads();
    PW_STAMP(2);
    // ---- multiply: blocks cb0 (and cb1) x the four tiles ----------------------------------------------------
    v4d acc[2][T];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[b][t] = v4d{0.0, 0.0, 0.0, 0.0};
    // TWO (does this wave carry a second block) is a COMPILE-TIME property of the loop: as a run-time flag it put a
    // branch around every second MFMA and every second fragment load, and hipcc's waitcnt pass, which must serve the
    // path with the fewest loads in flight, then waited for the fragments of k-block kb + 1 before the MFMAs of
    // k-block kb (vmcnt(7) .. vmcnt(0) with eight loads just issued): the prefetch distance was zero.
    // `tri_tag`: the fragment image holds the blocks cb >= kb only (mfma_pack_frags, tri = true); `member_tag`: which of
    // the wave's two blocks a one-block product works on; k-blocks [kb_begin, kb_end).
    auto product = [&](auto two_tag, auto member_tag, auto tri_tag, const double *__restrict__ frags, int kb_begin, int kb_end) {
      constexpr bool TWO = decltype(two_tag)::value;
      constexpr int MB = decltype(member_tag)::value;
      constexpr bool TRI = decltype(tri_tag)::value;
      if (kb_begin >= kb_end) return;
      // dense: fragment (kb, s, cb) is at ((kb 4 + s) NB + cb) x 64; triangular: k-block kb starts behind the
      // 4 (NB + NB - 1 + .. + NB - kb + 1) fragments of the earlier ones and holds 4 x (NB - kb).  Both operands of
      // k-block kb + 1 -- eight fragment values from L2, eight 16-byte pieces from LDS -- are requested before the
      // MFMAs of k-block kb are issued; two register sets swap roles (loop unrolled by two: no copies).
      double wa[4][2], wb[4][2];
      v2d xa[2][T], xb[2][T];
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(frags), 0,
                                                          (TRI ? 2 * NB * (NB + 1) : NB * 4 * NB) * 512, 0x00020000);
      const int lane8 = lane * 8;
      const int cbs = MB ? cb1 : cb0;  // the block of a one-block product
      auto load_w = [&](int kb, double(&dst)[4][2]) {
        const int kbase = TRI ? 4 * (kb * NB - kb * (kb - 1) / 2) - kb : 0, width = TRI ? NB - kb : NB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // (buffer loads: SGPR descriptor + scalar fragment offset + one shared lane offset, instead of a 64-bit
          // VGPR pointer per fragment -- a dozen such pointers were the difference between 256 VGPRs with spills
          // and none)
          const int row = TRI ? kbase + s * width : (kb * 4 + s) * NB;
          dst[s][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane8, (row + (TWO ? cb0 : cbs)) * 512, 0));
          if constexpr (TWO) dst[s][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane8, (row + cb1) * 512, 0));
        }
      };
      auto load_x = [&](int kb, v2d(&dst)[2][T]) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int t = 0; t < T; ++t) dst[h2][t] = reinterpret_cast<const v2d *>(sB + ((kb * 2 + h2) * T + t) * 128)[lane];
      };
      auto mfmas = [&](double(&wc)[4][2], v2d(&xc)[2][T]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const double bv = xc[s >> 1][t][s & 1];
            if constexpr (TWO) {
              acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s][0], bv, acc[0][t], 0, 0, 0);
              acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s][1], bv, acc[1][t], 0, 0, 0);
            } else {
              acc[MB][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s][0], bv, acc[MB][t], 0, 0, 0);
            }
          }
      };
      // The loop body requests UNCONDITIONALLY (the last one or two k-blocks are peeled off behind it): a branch
      // around a request leaves the waitcnt pass two paths to serve, and it serves the one without the new loads --
      // vmcnt(7) .. vmcnt(0) in front of MFMAs whose operands arrived a k-block ago, i.e. a wait for the loads
      // just issued on every second k-block.
      load_w(kb_begin, wa);
      load_x(kb_begin, xa);
      int kb = kb_begin;
#pragma unroll 1
      for (; kb + 2 < kb_end; kb += 2) {
        load_w(kb + 1, wb);
        load_x(kb + 1, xb);
        mfmas(wa, xa);
        load_w(kb + 2, wa);
        load_x(kb + 2, xa);
        mfmas(wb, xb);
      }
      if (kb + 2 == kb_end) {
        load_w(kb + 1, wb);
        load_x(kb + 1, xb);
        mfmas(wa, xa);
        mfmas(wb, xb);
      } else {
        mfmas(wa, xa);
      }
    };
    using Two = std::true_type;
    using One = std::false_type;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    auto multiply_dense = [&](const double *__restrict__ frags) {
      if (NB == 16 || two) product(Two{}, M0{}, std::false_type{}, frags, 0, NB);
      else product(One{}, M0{}, std::false_type{}, frags, 0, NB);
    };
    // triangular: both blocks up to k-block cb0, then the upper block alone up to cb1; a one-block wave up to cb0
    auto multiply_tri = [&](const double *__restrict__ frags) {
      if (NB == 16 || two) {
        product(Two{}, M0{}, std::true_type{}, frags, 0, cb0 + 1);
        product(One{}, M1{}, std::true_type{}, frags, cb0 + 1, cb1 + 1);
      } else {
        product(One{}, M0{}, std::true_type{}, frags, 0, cb0 + 1);
      }
    };
    if constexpr (TRIQ) multiply_tri(fragsQ);
    else multiply_dense(fragsQ);
    PW_STAMP(3);
    // lane (p, h), register r of block b holds output dim j = 16 cb_b + h + 4 r of particle p
    if constexpr (MVT) {
      // Student-t: Q xi scaled per component by sqrt(nu / chi2) BEFORE G x_prev is added on top.  A batch of draws
      // wants ~150 registers; with the 64 accumulator registers live beside it hipcc spilled the accumulators to
      // scratch and picked them out again piece by piece (~6 GB of scratch traffic per 5e5 x 256 launch).  The slabs
      // are free at this point and are exactly 8 waves x 16 KB: each wave parks its accumulators in its own 16 KB,
      // scales them there one tile per trip of a ROLLED loop (one copy of the batch in the instruction stream
      // instead of four), and takes them back.
      __syncthreads();  // every wave has read the normals
      v2d *park = reinterpret_cast<v2d *>(sB + w * 2048) + lane;  // plane ((b T + t) 2 + i): registers 2 i, 2 i + 1
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t) {
          park[((b * T + t) * 2 + 0) * 64] = v2d{acc[b][t][0], acc[b][t][1]};
          park[((b * T + t) * 2 + 1) * 64] = v2d{acc[b][t][2], acc[b][t][3]};
        }
#pragma unroll 1
      for (int t = 0; t < T; ++t) {
        const long local = base + 16 * t + p;
        const uint32_t gi = first + (uint32_t)(local < (long)count ? local : (long)count - 1);
        double chi[8];
        chi_square_clayout<2>(cs, gi, step, k0, k1, h, [&](int b) { return 16 * (b ? cb1 : cb0); },
                              [&](int b, int j) { return (b == 0 || two) && (!PAD || j < d); }, chi, chi_q);
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          v2d *q = park + (((c >> 2) * T + t) * 2 + ((c >> 1) & 1)) * 64;
          const v2d v = *q;
          *q = v2d{v[0] * sqrt((double)nu / chi[c]), v[1] * sqrt((double)nu / chi[c + 1])};
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const v2d lo = park[((b * T + t) * 2 + 0) * 64], hi = park[((b * T + t) * 2 + 1) * 64];
          acc[b][t] = v4d{lo[0], lo[1], hi[0], hi[1]};
        }
    }
    if constexpr (HAS_G) {
      if constexpr (MVT) gather();
      __syncthreads();  // every wave has read the normals: the slabs take the gathered rows
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) reinterpret_cast<v2d *>(sB + (w + 8 * kb) * 128)[lane] = xg[kb];
      PW_STAMP(4);
      __syncthreads();
      multiply_dense(tail);  // (G is a general matrix)
    }
    PW_STAMP(5);
    if constexpr (HAS_G && !MVT) {
      if (g + gridDim.x < num_groups) anc_ahead = ancestor_of(g + gridDim.x);  // (ahead of the stores below)
    }
    // ---- epilogue ------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const long local = base + 16 * t + p;
      if (local >= (long)count) continue;
      const uint32_t gi = first + (uint32_t)local;
      const double *xrow = nullptr;
      if constexpr (MODE == 4) xrow = X_prev + (long)(a ? a[local] : gi) * d;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b == 1 && !two) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * (b ? cb1 : cb0) + h + 4 * r;
          if (PAD && j >= d) continue;
          double v = acc[b][t][r];
          if constexpr (MODE == 0) v += tail[j];
          if constexpr (MODE == 4) v += fma(tail[j], xrow[j], 0.0);  // (the one non-zero term of the dense kernels' sum)
          X_out[local * d + j] = v;
        }
      }
    }
    PW_STAMP(6);
    __syncthreads();  // every wave has read the slabs: the next group may overwrite them
    PW_STAMP(7);
  }
}

template <int NB, bool MVT, int MODE, bool PAD, bool TRIQ>
static hipError_t launch_pw(float nu, const double *X_prev, const uint32_t *a, const double *fragsQ, const double *tail,
                            int d, double scale, uint64_t seed, uint32_t step, uint32_t domain, uint32_t first,
                            uint32_t count, double *X_out, int num_cus, hipStream_t stream)
{
  const size_t lds_bytes = propagate_wide_lds_bytes(NB, MVT);
  auto kern = propagate_wide_kernel<NB, MVT, MODE, PAD, TRIQ>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  const long num_groups = ((long)count + kWideGroup - 1) / kWideGroup;
  long blocks = num_cus;
  if (blocks > num_groups) blocks = num_groups;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, stream, nu, X_prev, a, fragsQ, tail, d, scale,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, domain, first, count, X_out, num_groups);
  return hipGetLastError();
}

// fragsQ: dense mfma_pack_frags image of Q zero-padded to 16 NB (launch_propagate_mfma_wide_tri: Q lower triangular,
// packed with tri = true); tail: the dense image of G (mode 1), diag(G) (mode 4) or m0 (mode 0), both padded to 16 NB
// entries where they are vectors.
#if CUSMC_TRIQ
#define CUSMC_PW_ENTRY launch_propagate_mfma_wide_tri
#else
#define CUSMC_PW_ENTRY launch_propagate_mfma_wide
#endif
hipError_t CUSMC_PW_ENTRY(int kind, float nu, const double *X_prev, const uint32_t *a, const double *fragsQ,
                                      const double *tail, int mode, int d, double scale, uint64_t seed, uint32_t step,
                                      uint32_t domain, uint32_t first, uint32_t count, double *X_out, int num_cus,
                                      hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  const bool mvt = kind == CUSMC_MVT;
  const bool pad = d % 16 != 0;
  constexpr bool TQ = CUSMC_TRIQ != 0;
#define CUSMC_PW_MODE(nb, m)                                                                                              \
  (mvt ? (pad ? launch_pw<nb, true, m, true, TQ>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)   \
              : launch_pw<nb, true, m, false, TQ>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)) \
       : (pad ? launch_pw<nb, false, m, true, TQ>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)  \
              : launch_pw<nb, false, m, false, TQ>(nu, X_prev, a, fragsQ, tail, d, scale, seed, step, domain, first, count, X_out, num_cus, stream)))
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
#undef CUSMC_PW_ENTRY
  return hipErrorInvalidValue;
}

}  // namespace cusmc
