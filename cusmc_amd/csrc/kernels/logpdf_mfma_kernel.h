// Batched multivariate Normal / Student-t log-density for 16 <= d <= 176 (NB = ceil(d/16) <= 11 blocks) on
// gfx950, fp64.
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
//   B operand  lane (p = lane&15, h = lane>>4) holds r[p][k], k = 16*kb + pi(s,h), for k-step s
//              of k-block kb, where pi(s,h) = 2h + (s&1) + 8(s>>1).  The k order inside a block
//              is free (it is a summation index) and this one lets each lane fetch its four
//              values of a block with two 16-byte loads, 64 contiguous bytes per particle per
//              load instruction: X goes HBM -> VGPR once, coalesced, with no LDS round trip.
//   A operand  lane (j = lane&15, h) holds M[16*cb + j][16*kb + pi(s,h)]: the factor, packed on
//              the host in exactly this order (mfma_pack_frags), staged ONCE per workgroup in LDS
//              and from there kept in registers for the whole kernel (d <= 64: 80 VGPRs at d = 64)
//              or, for the larger forms, read from LDS one k-step ahead of its use.
//   C/D        C = M R^T: lane (p, h), register r  ->  output dim 16*cb + h + 4r of PARTICLE p.
//              A lane only ever holds one particle's outputs, so the row sum of squares is 16
//              in-lane FMAs plus one 4-lane reduction over h -- not a 16-lane reduction of four
//              values, which is what the transposed product costs.
//   Epilogue   lanes 0..15 finish particles 0..15 of the tile and store one 128-byte line (Student-t:
//              once per four tiles, on all 64 lanes).
//
// Roofline: 8d + 8 algorithmic bytes per particle; at d = 64 the kernel needs 40
// MFMAs of 2048 flop per 16 particles.
#pragma once
#include <hip/hip_runtime.h>

#include "smallops.h"

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));  // rows are 8-byte aligned in general

__host__ __device__ constexpr int pi_k(int s, int h) { return 2 * h + (s & 1) + 8 * (s >> 1); }

__device__ __forceinline__ double finish(double q, const Epilogue &ep)
{
  double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p_nonneg(q * ep.inv_nu)
                                     : ep.lognorm - 0.5 * q;
  return ep.out_density ? exp(lp) : lp;
}

template <int EPI>
__device__ __forceinline__ double finish_epi(double q, const Epilogue &ep)
{
  if (EPI == 1) return ep.lognorm - 0.5 * q;
  if (EPI == 2) return ep.lognorm - ep.half_nu_plus_d * log1p_nonneg(q * ep.inv_nu);
  return finish(q, ep);
}

// CENTRED  = centred form:  z = W (x - shift), no bias                       (pdf(y, F))
// !CENTRED = affine form:   z = bias + L x,    no shift                      (reweight_G)
// The factor is lower triangular in BOTH forms.  reweight_G's z = W y - (W F) x has a dense matrix
// M = -W F, but only |z|^2 is wanted and that is invariant under rotations of z: the host factors
// M = Q L (Householder QL, hostla.h) and hands the kernel L and the rotated bias Q^T W y, so a
// general observation matrix F costs the same 40 block-products per tile at d = 64 as F = I does,
// not the 64 of the dense product (133 -> 9x us for 1e6 particles).  The bias is
// the initial accumulator of each output block's first MFMA: it costs no instruction at all.
// SHIFT = false drops the subtraction when the shift vector is all zeros.
// EPI fixes the epilogue at compile time: 1 = MVN log-density (lognorm - q/2), 2 = Student-t
// log-density (lognorm - (nu+d)/2 log1p(q/nu), log1p through the library's own ln: smallops.h),
// 0 = decided at run time (densities: exp on top).  The
// run-time epilogue (Student-t log1p, optional exp) costs ~30 VGPRs of polynomial constants that
// stay live across the tile loop, and at d = 64 the loop owns the whole register file (80 factor +
// 96 operand + 32 accumulator VGPRs) -- with them hipcc spilled one operand pair and drained
// vmcnt to reload it on every third tile.
// PAD = true serves every other d in (16, 176] and every alignment: the factor is zero-padded to
// 16*NB on the host, the first NB-1 k-blocks are loaded as usual (16-byte loads that need no more
// than the rows' natural 8-byte alignment), and the last k-block is loaded element by element with
// the column clamped into the row (columns >= d read the particle's own last element, against a zero
// factor column) -- never from the next row (whose values are not this particle's and might not be
// finite) nor past the end of X.
// ABL is for scripts/calib/ablate.hip only (0 in the library): 1 = no global loads inside the tile
// loop, 2 = no MFMAs, 3 = no cross-lane reduction, 4 = clock stamps, 5 = 1 + 4.  It exists to
// attribute time.
//
// Instruction budget.  Measured on gfx950 (scripts/calib/calib.hip): while a v_mfma_f64 runs
// (64.8 cycles) NO other wave of that SIMD issues VALU work -- an MFMA wave plus an integer-VALU
// wave take the SUM of their times.  Kernel time per SIMD is therefore (MFMA cycles) + (every
// VALU instruction of every wave), whatever the occupancy, and each tile's 40 MFMAs (2592
// cycles) leave room for only a few dozen VALU instructions before the kernel stops being
// HBM-bound.  Hence: tile addresses are scalar (SGPR base + a loop-invariant 32-bit lane offset),
// accumulators stay in VGPRs (the library is built with -amdgpu-mfma-vgpr-form, so the epilogue
// needs no v_accvgpr_read), one accumulator per output block, and the epilogue is 16 FMAs, two
// adds and one compare per tile.
//
// Work distribution.  ONE workgroup per CU, as many waves as the register file admits
// (mfma_threads<NB>()).
//   * Tiles are dealt to workgroups round-robin: workgroup b owns tiles b, b + G, b + 2G, ...  At
//     any moment the G workgroups stream one contiguous ~2 MB window of X.  Handing each workgroup
//     chunks of 16 CONSECUTIVE tiles from a global queue instead -- round 1's first scheduler --
//     measured 2 - 4.5 us per launch slower (same box, same build, scripts/calib/ablate.hip): each
//     chunk request cost its wave a full s_waitcnt vmcnt(0), i.e. the HBM latency under load with
//     nothing prefetched behind it, and 256 scattered 128 KB streams use HBM less well than one
//     window.
//   * Inside a workgroup the waves pull their next ROUND from one LDS counter (round k of workgroup b
//     is tile b + k G) -- a single ds_add_rtn_u32 per tile.  Static round-robin over WAVES loses ~10 %:
//     the older of two waves on a SIMD wins issue arbitration, finishes its share early and leaves
//     the younger one to run alone (measured: 78 us vs 87 us wave lifetimes at d = 64).
//   * Nothing is dynamic ACROSS workgroups, although workgroup lifetimes differ by ~8 % from launch
//     to launch -- at random: the slow ones are not the same from one launch to the next (rank
//     correlation 0.07, scripts/calib/ablate.hip persist) -- and the launch ends ~5 us after its average
//     workgroup.  Two queued tails were built
//     and measured against the pure deal on one box each (scripts/calib/ablate.hip):
//       - last eighth of the rounds from ONE counter, chunks of 16 then 8 tiles, scalar
//         s_atomic_add ... glc draws (they return through lgkmcnt, so the drawing wave keeps its
//         vector loads in flight; gfx950 runs scalar atomics coherently across XCDs at ~87 per us
//         per word, scripts/calib/satomic.hip), a 64-bit LDS word per tile:        +1.0 us;
//       - last eighth from EIGHT counters (slice b % 8: ~11 draws per us per word), chunks of 8
//         (one tile per wave) drawn a chunk ahead, dealt rounds on the unchanged cheap path:
//         +3.7 us -- and +5.2 us with the tail compiled in but switched off at run time.  The
//         balancing itself gains ~1.5 us; the extra paths through grab() cost the tile loop its
//         load placement and counted waits (an s_waitcnt vmcnt(0) every second tile).
//     A tail worth having would have to live in a loop of its own, and then pays a pipeline
//     drain per wave at the hand-over: not pursued.
//
// Factor residency.  The factor is staged once per workgroup in LDS.  WREG (d <= 64): each lane
// then copies its 2*NB*(NB+1) factor values into registers for the whole kernel (80 VGPRs at
// d = 64) -- the tile loop has no LDS reads at all.  Otherwise the fragments stay in LDS and are
// read one k-step ahead.
//
// Things that were tried on the prologue and the loop shape and measured SLOWER (ablate.hip quick
// mode, same box): touching the first tiles' cache lines before the factor loads (+2.3 us: 64
// distinct lines per instruction swamp the L1 miss queue and the factor loads sit behind them);
// a single-exit tile loop with guarded computes, which gives textbook counted waits (+1.6 us);
// non-temporal output stores (+0.3 us) or particle loads (+1.4 us); spreading the next tile's loads
// between the MFMAs with sched_group_barrier (+0.7 us); rotating each workgroup's slot from round to round, in case a
// fixed slot pinned a workgroup to the same HBM channels (no effect: 94.4-95.2 us for rotations 0,
// 1, 8, 37, 97 on one box); giving every wave its first round by birth and an LDS-only barrier, so
// that the factor loads and the first tile's loads are in flight together (+0.6 us: 96.0 -> 96.6
// median over three interleaved runs, although the load-free variant gains 1 us) -- and again after
// the factor started going through LDS, this time with the first tile's loads really in flight
// across the barrier (s_waitcnt vmcnt(8) before the LDS writes, lgkmcnt(0) only before s_barrier):
// 95.4 -> 95.2 us, inside the noise, not kept.
template <int NB>
__host__ __device__ constexpr int mfma_threads()
{
#ifdef EXP_THREADS4  // calibration builds only (scripts/calib/ablate.hip)
  if (NB == 4) return EXP_THREADS4;
#endif
  // (512 threads for NB >= 5 fits with 12-100 bytes of spill and measured within +-3 % of 256,
  // better at d = 80, worse at d = 112 and for the Student-t epilogue: not worth the spills)
  return NB == 1 ? 1024 : NB == 2 ? 768 : NB <= 4 ? 512 : 256;
}
template <int NB>
__host__ __device__ constexpr bool mfma_factor_in_regs()
{
#ifdef EXP_LDSW
  return false;
#else
  return NB <= 4;
#endif
}
template <int NB, bool SHIFT, int EPI>
__host__ __device__ constexpr int mfma_sets()  // operand register sets in rotation
{
#ifdef EXP_SETS
  return NB <= 4 ? EXP_SETS : 2;
#elif defined(EXP_SETS_BIG)  // calibration: a third operand set for the LDS-factor variants too
  return NB <= EXP_SETS_BIG ? 3 : 2;
#else
  // Two sets (one tile in flight behind the one being multiplied) beat three at every d <= 64
  // without a shift: 91.2 vs 94.6 us at d = 64, 93.2 vs 95.1 at d = 32 (interleaved runs on one
  // box, scripts/logpdf_sweep.py).  2048 waves x 2 tiles is already 32 MB of X in flight, more
  // than the HBM latency-bandwidth product needs, and a wider window of open addresses costs DRAM
  // locality.  The shifted form at d >= 48 is the exception (92.6 vs 98.7 us at d = 48): its
  // per-step subtractions lengthen a tile enough for the deeper prefetch to pay.  d >= 96 has no
  // registers for a third set.
  // With the run-time epilogue on top (Student-t, densities) three sets plus the hoisted shift values
  // spill at d = 64 (128 us instead of 107), so those stay at two.
  return SHIFT && EPI == 1 && (NB == 3 || NB == 4) ? 3 : 2;
#endif
}

template <int NB, bool CENTRED, bool SHIFT, int ABL = 0, int EPI = 0, bool PAD = false>
__global__ __launch_bounds__(mfma_threads<NB>()) void logpdf_mfma_kernel(
    const double *__restrict__ X, long N, long ldx, const double *__restrict__ frags,
    const double *__restrict__ shift, const double *__restrict__ bias, Epilogue ep,
    double *__restrict__ out, long num_tiles, int d_true = 16 * NB)
{
  constexpr int THREADS = mfma_threads<NB>();
  constexpr bool STAMP = ABL == 4 || ABL == 5;
  constexpr bool NOLOAD = ABL == 1 || ABL == 5;
  unsigned long long stamp_entry = 0;
  if (STAMP) stamp_entry = __builtin_amdgcn_s_memrealtime();
  constexpr int NFRAG = 4 * NB * (NB + 1) / 2;
  constexpr bool WREG = mfma_factor_in_regs<NB>();
  extern __shared__ double lds[];
  double *sShift = lds;              // 16*NB
  double *sBias = sShift + 16 * NB;  // 16*NB
  // sNext: the workgroup's round counter (tile b + k*G is round k), one ds_add_rtn_u32 per tile.
  // (a relaxed workgroup-scope atomic, not `volatile`: volatile LDS accesses make the memory
  // legaliser drain vmcnt as well, which costs the tile loop its counted waits)
  unsigned *sNext = reinterpret_cast<unsigned *>(sBias + 16 * NB);
  double *sF = sBias + 16 * NB + 4;  // NFRAG x 64 (WREG: the prologue's broadcast buffer only)

  {
    // Stage the factor in LDS: all of a chunk's 16-byte loads are issued before the first LDS write,
    // so the prologue costs one memory round trip per chunk, not one per element.  The register-
    // resident variants (WREG) go through LDS as well and pick their fragments up after the barrier:
    // the factor then crosses L2 -> CU once per workgroup instead of once per wave (8 x 20 KB at
    // d = 64, 41 MB chip-wide in the first microseconds of every launch): -0.9 us per launch,
    // 94.1 -> 93.2 us median over three interleaved runs (scripts/calib/ablate.hip).
    constexpr int NV = NFRAG * 32;  // 16-byte elements
    constexpr int SC = 8;
    const v2d *g = reinterpret_cast<const v2d *>(frags);
    v2d *l = reinterpret_cast<v2d *>(sF);
    for (int base = 0; base < NV; base += SC * THREADS) {
      v2d tmp[SC];
#pragma unroll
      for (int c = 0; c < SC; ++c) {
        const int i = base + c * THREADS + (int)threadIdx.x;
        if (i < NV) tmp[c] = g[i];
      }
#pragma unroll
      for (int c = 0; c < SC; ++c) {
        const int i = base + c * THREADS + (int)threadIdx.x;
        if (i < NV) l[i] = tmp[c];
      }
    }
  }
  if ((SHIFT || !CENTRED) && threadIdx.x < 16 * NB) {  // (a cold miss in front of the barrier otherwise)
    sShift[threadIdx.x] = shift[threadIdx.x];
    sBias[threadIdx.x] = bias[threadIdx.x];
  }

  const unsigned G = gridDim.x;
  const int lane = threadIdx.x & 63;
  const unsigned nt = (unsigned)num_tiles;  // launch_nb() guarantees num_tiles < 2^31
  const unsigned rounds = (nt + G - 1) / G;
  if (threadIdx.x == 0) *sNext = 0;

  const int p = lane & 15, h = lane >> 4;
  const long last = num_tiles - 1;
  const long tile_bytes = 128 * ldx;  // 16 rows
  // per-lane byte offset inside a tile; in the last tile rows past N re-read row N-1 (their
  // stores are masked).  mfma_supported() guarantees 16*ldx*8 < 2^32.
  const long tail_rows = N - last * 16;  // 1..16
  const unsigned lane_off = (unsigned)((long)p * ldx + 2 * h) * 8u;
  const unsigned lane_off_last = (unsigned)((long)(p < tail_rows ? p : tail_rows - 1) * ldx + 2 * h) * 8u;

  // Next tile for this wave (wave-uniform); nt means the workgroup's share is exhausted.  Scalar
  // 32-bit arithmetic (64-bit scalar compares do not exist and would fall back to the VALU).
  auto grab = [&]() -> unsigned {
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(sNext, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    const unsigned t = blockIdx.x + k * G;  // k < rounds: no overflow (nt < 2^31, G <= num_cus)
    return k < rounds && t < nt ? t : nt;
  };

  // loads tile tt; past the end it re-reads the workgroup's first tile (an L2 hit, result
  // unused): a branch around the prefetch would make hipcc's s_waitcnt placement assume the
  // no-prefetch path and wait for the prefetched loads at the head of every tile.
  // PAD: per-lane byte offsets of the last k-block's four elements (k = pi(s,h)), column clamped to
  // d-1, for a full tile and for the last one
  unsigned pad_off[4] = {0, 0, 0, 0}, pad_off_last[4] = {0, 0, 0, 0};
  if constexpr (PAD) {
    const int rem = d_true - 16 * (NB - 1);  // columns of the last block that exist: 1..16
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int col = pi_k(s4, h);
      const long cc = 16 * (NB - 1) + (col < rem ? col : rem - 1);
      pad_off[s4] = (unsigned)((long)p * ldx + cc) * 8u;
      pad_off_last[s4] = (unsigned)((long)(p < tail_rows ? p : tail_rows - 1) * ldx + cc) * 8u;
    }
  }
  auto load_tile = [&](unsigned tt, v2d(&a)[NB][2]) {
    const long t = tt < nt ? tt : blockIdx.x;
    const char *base = reinterpret_cast<const char *>(X) + t * tile_bytes;  // scalar
    const unsigned off = t == last ? lane_off_last : lane_off;
#pragma unroll
    for (int kb = 0; kb < (PAD ? NB - 1 : NB); ++kb) {
      a[kb][0] = *reinterpret_cast<const v2d_a8 *>(base + off + 128 * kb);
      a[kb][1] = *reinterpret_cast<const v2d_a8 *>(base + off + 128 * kb + 64);
    }
    if constexpr (PAD) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        // A column >= d reads THIS particle's own last element instead (clamped address) and meets a
        // zero column of the padded factor: 0 for every finite particle, and a particle holding a NaN /
        // Inf there is non-finite in the exact result too.  No select -- on the loaded value it made
        // the wave wait for these loads, the youngest in flight, the moment they were issued, and with
        // them for the whole prefetched tile (7 - 18 % of a launch).
        a[NB - 1][s4 >> 1][s4 & 1] = *reinterpret_cast<const double *>(base + (t == last ? pad_off_last[s4] : pad_off[s4]));
      }
    }
  };

  __syncthreads();
  double wreg[WREG ? NFRAG : 1];
  if (WREG) {
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) wreg[f] = sF[f * 64 + lane];
  }
  // the shift values a lane subtracts are the same for every tile: 4*NB registers instead of
  // 4*NB LDS reads per tile (d <= 64; above that the registers are spoken for)
  double shreg[(CENTRED && SHIFT && WREG) ? NB : 1][4];
  if constexpr (CENTRED && SHIFT && WREG) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) shreg[kb][s4] = sShift[16 * kb + pi_k(s4, h)];
  }
  // likewise the affine form's bias: C rows of block cb are output dims 16 cb + h + 4r
  v4d breg[(!CENTRED && WREG) ? NB : 1];
  if constexpr (!CENTRED && WREG) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const double *b = sBias + 16 * cb + h;
      breg[cb] = v4d{b[0], b[4], b[8], b[12]};
    }
  }
  // (!WREG) the factor fragments are loop-invariant LDS reads; left alone, hipcc hoists all of
  // them out of the tile loop into spilled registers.  An opaque per-tile lane offset keeps them
  // as in-loop ds_read_b64.
  int lds_lane = lane;

  // Student-t epilogue, batched (EPI == 2 unless EXP_NOBATCH): after the 4-lane reduction every lane
  // (p, h) holds particle p's q, so the q of four consecutive tiles are parked in the four h-groups
  // (one per-lane select each) and the log1p epilogue runs once per four tiles on all 64 lanes --
  // its ~40 f64 VALU instructions are serialised with the MFMAs, a quarter as often.
#ifdef EXP_NOBATCH
  constexpr bool BATCH = false;
#else
  constexpr bool BATCH = EPI == 2;
#endif
  double q_parked = 0.0;
  long row_parked = N;  // N = nothing parked
  unsigned parked = 0;  // tiles computed by this wave (wave-uniform)
  auto flush = [&]() {
    if (row_parked < N) out[row_parked] = finish_epi<EPI>(q_parked, ep);
    row_parked = N;
  };
  auto compute_tile = [&](unsigned tu, const v2d(&a_in)[NB][2]) {
    const long t = tu;
    v4d acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      if (CENTRED) {
        acc[cb] = v4d{0.0, 0.0, 0.0, 0.0};
      } else if (WREG) {
        acc[cb] = breg[cb];
      } else {  // C rows are output dims h + 4r of block cb
        const double *b = sBias + 16 * cb + h;
        acc[cb] = v4d{b[0], b[4], b[8], b[12]};
      }
    }
    int f = 0;
    if constexpr (WREG) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          double a = a_in[kb][s >> 1][s & 1];
          if (CENTRED && SHIFT) a -= shreg[kb][s];
#pragma unroll
          for (int cb = kb; cb < NB; ++cb, ++f) {
            if (ABL == 2)
              acc[cb][0] += a + wreg[f];  // keeps loads and factor live without the matrix pipe
            else  // A = factor rows (output dims), B = particles: C[row = out dim][col = particle]
              acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(wreg[f], a, acc[cb], 0, 0, 0);
          }
        }
      }
    } else {
      asm volatile("" : "+v"(lds_lane));
      // the factor fragment of step n+1 is read from LDS before the MFMAs of step n are issued
      double w_cur[NB], w_nxt[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) w_cur[cb] = sF[(f + cb) * 64 + lds_lane];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int lo = kb;
          const int cnt = NB - lo;
          const int kbn = s == 3 ? kb + 1 : kb;  // the next step's k-block
          [[maybe_unused]] const int lon = kbn;
#ifndef EXP_NOLDSW  // (calibration builds: the factor fragments are read once per tile -- wrong results, attributes the LDS waits)
          if (kbn < NB) {
#pragma unroll
            for (int cb = lon; cb < NB; ++cb) w_nxt[cb] = sF[(f + cnt + cb - lon) * 64 + lds_lane];
          }
#endif
          double a = a_in[kb][s >> 1][s & 1];
          if (CENTRED && SHIFT) a -= sShift[16 * kb + pi_k(s, h)];
#pragma unroll
          for (int cb = lo; cb < NB; ++cb) {
            if (ABL == 2)
              acc[cb][0] += a + w_cur[cb];
            else
              acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(w_cur[cb], a, acc[cb], 0, 0, 0);
          }
          f += cnt;
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) w_cur[cb] = w_nxt[cb];
        }
      }
    }
    // lane (p, h) holds output dims {h + 4r} of every block for particle p: square-sum in the
    // lane, then one 4-lane reduction over h (lanes p, p+16, p+32, p+48).
    double q = 0.0;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) q = fma(acc[cb][r], acc[cb][r], q);
    }
    if (ABL != 3) {
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
    }
    if constexpr (BATCH) {
      const int slot = (int)(parked & 3u);
      ++parked;
      const bool mine = h == slot;
      q_parked = mine ? q : q_parked;
      row_parked = mine ? t * 16 + p : row_parked;  // (rows >= N of the last tile park as "nothing": flush() checks)
      if (slot == 3) flush();
    } else {
      // lanes 0..15: particles 0..15 of the tile, one 128-byte line
      if (lane < (t == last ? (int)tail_rows : 16)) out[t * 16 + lane] = finish_epi<EPI>(q, ep);
    }
  };

  // ABL == 4 (diagnostic build only): shader-clock and 100 MHz wall stamps per wave, written past
  // the end of `out`, never mixed into an output value.
  unsigned long long stamp_c = 0, stamp_r = 0;
  if (STAMP) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  int done = 0;

  unsigned k0 = grab();
  if (k0 < nt) {
    if constexpr (mfma_sets<NB, SHIFT, EPI>() == 3) {
      // Three register sets in rotation: while one tile runs on the matrix cores the loads of
      // the next TWO are in flight (16 KB per wave).  No register copies: the loop is unrolled
      // by three with the roles renamed.
      v2d a0[NB][2], a1[NB][2], a2[NB][2];
      unsigned k1 = grab();
      load_tile(k0, a0);
      load_tile(k1, a1);
      while (true) {
        const unsigned k2 = grab();
        if (!NOLOAD) load_tile(k2, a2);
        compute_tile(k0, a0); ++done;
        if (k1 >= nt) break;
        k0 = grab();
        if (!NOLOAD) load_tile(k0, a0);
        compute_tile(k1, NOLOAD ? a0 : a1); ++done;
        if (k2 >= nt) break;
        k1 = grab();
        if (!NOLOAD) load_tile(k1, a1);
        compute_tile(k2, NOLOAD ? a0 : a2); ++done;
        if (k0 >= nt) break;
      }
    } else {  // two sets in rotation
      v2d a0[NB][2], a1[NB][2];
      load_tile(k0, a0);
      while (true) {
        const unsigned k1 = grab();
        if (!NOLOAD) load_tile(k1, a1);
        compute_tile(k0, a0); ++done;
        if (k1 >= nt) break;
        k0 = grab();
        if (!NOLOAD) load_tile(k0, a0);
        compute_tile(k1, NOLOAD ? a0 : a1); ++done;
        if (k0 >= nt) break;
      }
    }
  }
  if constexpr (BATCH) flush();
  if (STAMP && lane == 0) {
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(out + num_tiles * 16);
    const long wid = (long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    dbg[3 * wid] = __builtin_amdgcn_s_memtime() - stamp_c;
    dbg[3 * wid + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r;
    dbg[3 * wid + 2] = (unsigned long long)done;
    unsigned long long *abs_t = dbg + 3 * (long)gridDim.x * (THREADS / 64);  // absolute 100 MHz stamps
    abs_t[3 * wid] = stamp_entry;
    abs_t[3 * wid + 1] = stamp_r;
    abs_t[3 * wid + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace cusmc
