// Batched MVN / Student-t log-density for large d: 176 < d <= 256, run as NB = ceil(d / 16) = 12 .. 16 blocks of 16:
// the regime where (X - mu) L^-T is a genuine dense GEMM and the kernel is bound by the f64 matrix
// cores, not by HBM (d = 256: 2056 B and ~70 kflop per particle = 34 flop/B against a machine
// balance of 9.8).  Same contract and same reference functions as kernels/logpdf_mfma_kernel.h:
//     z = L (x - shift)  or  z = bias + L x,  L lower triangular,  q = z.z,  out = epilogue(q)
//
// Why a second kernel.  At d = 256 the block-triangular factor is 278 KB: it fits neither the
// register file nor LDS, and one wave cannot hold 16 output blocks of accumulators for several
// particle tiles.  So the OUTPUT dimension is split over the waves of a workgroup:
//   * one workgroup per CU; it walks groups of GP particles (32; 48 at NB = 12, see wide_gp());
//   * a wave owns the PAIR of output blocks (q, NB-1-q) for the group's two particle tiles (NB = 12:
//     four pairs and four single blocks -- see wide_waves()).  In the triangular form block cb needs
//     k-blocks 0..cb, so every pair costs (q+1) + (NB-q) = NB+1 block-products: all waves carry
//     exactly the same number of MFMAs (136 / 8 = 17 at d = 256);
//   * accumulators are 2 tiles x 2 blocks x 8 = 32 VGPRs, which leaves the register file free
//     for software pipelining.  That matters more than anything else here: an f64 MFMA blocks
//     VALU issue on its SIMD and a wave is in-order, so every operand must already be in a
//     register when its MFMA is due.  Earlier versions of this kernel ran at
//     256 VGPRs and hipcc sank every prefetch next to its use: 35-40 % of the time both waves of
//     a SIMD sat in s_waitcnt (scripts/calib/ablate_wide.cpp);
//   * particle rows: staged ONCE per group in LDS in MFMA operand order
//     ([k-block][half][tile][lane][2]: one conflict-free ds_read_b128 per two operands), double
//     buffered, and filled by a DEDICATED LOADER WAVE (the last wave of the workgroup; LDS-DMA,
//     so no VGPR or VALU traffic beside the compute waves of its SIMD) while the
//     compute waves work on the other buffer -- a whole group (~8 us) of prefetch distance.  The
//     HBM stream needs its own wave because vmcnt retires in order: any wave that has an HBM
//     miss in flight makes its later, L2-resident fragment loads wait behind it (measured: the
//     first fragment wait of every group stalled for the full HBM latency);
//   * factor fragments: streamed from L2 with buffer loads (SGPR descriptor + lane offset; flat
//     loads make hipcc materialise dozens of 64-bit pointers), each used by two MFMAs, fetched a
//     whole k-block ahead into the idle one of two register sets (k loop unrolled by two: no
//     register copies);
//   * C has particles on columns: a lane's accumulators belong to one particle, so the partial
//     sum of squares is in-lane FMAs + one 4-lane reduction; the waves' partials meet in LDS
//     once per group (the only barrier) and are added in a fixed order.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../launch.h"
#include "../../../include/cusmc_hip.h"
#include "smallops.h"

namespace cusmc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// EIGHT compute waves (two per SIMD: waves w and w + 4) whatever the block count NB = 9 .. 16, each
// carrying up to two output blocks (lo < hi; lo = -1: a single block; hi = -1: none) for BOTH particle
// tiles of a group, so that every fragment it loads feeds two MFMAs.  Output block cb costs cb + 1
// block-products per tile; the blocks are dealt to the four SIMDs longest-first onto the least loaded
// one (at most four per SIMD), and a SIMD's blocks to its two waves as (largest, smallest) and the
// middle two.  NB = 16 comes out as 34 block-products per tile on every SIMD, NB = 12 as 21, 20, 19,
// 18 where 19.5 would be perfect (six pair-waves would leave 26, 26, 13, 13: the first NB = 12 mapping,
// 48.8 TFLOP/s at d = 192).
__host__ __device__ constexpr int wide_waves(int nb) { return (void)nb, 8; }
// particles per group: 32, and 48 at NB = 12, the one block count whose two staging buffers of three tiles fit
// the 160 KB of LDS (147 KB): a group costs ~2000 cycles that are not MFMAs whatever its size, so a tile more per
// barrier is worth 68 -> 7x % matrix-core utilisation at d = 177 .. 192
__host__ __device__ constexpr int wide_gp(int nb) { return nb == 12 ? 48 : 32; }
struct WideMap { int lo[8], hi[8]; };
__host__ __device__ constexpr WideMap wide_map(int nb)
{
  WideMap m{};
  for (int w = 0; w < 8; ++w) { m.lo[w] = -1; m.hi[w] = -1; }
  int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0}, blk[4][4] = {};
  for (int cb = nb - 1; cb >= 0; --cb) {  // longest first
    int s = -1;
    for (int k = 0; k < 4; ++k)
      if (cnt[k] < 4 && (s < 0 || load[k] < load[s])) s = k;
    blk[s][cnt[s]++] = cb;  // (descending within a SIMD)
    load[s] += cb + 1;
  }
  for (int s = 0; s < 4; ++s) {
    const int n = cnt[s];
    if (n >= 1) m.hi[s] = blk[s][0];
    if (n >= 2) m.hi[s + 4] = blk[s][1];
    if (n == 3) m.lo[s] = blk[s][2];
    if (n == 4) { m.lo[s] = blk[s][3]; m.lo[s + 4] = blk[s][2]; }
  }
  return m;
}
// The wave that adds up a group's partial sums and runs the epilogue: the older wave (w < 4) of the LEAST loaded SIMD.
// Its ~20 (Normal) .. ~80 (Student-t) dependent VALU instructions issue one per gap between the other waves' 64-cycle
// MFMAs -- thousands of cycles -- and on the most loaded SIMD (wave 0's, where the map puts the largest block) that
// lands on the group's critical path: Student-t at NB = 12 cost 22 % more than Normal with wave 0 doing it.
__host__ __device__ constexpr int wide_epilogue_wave(int nb)
{
#ifdef CUSMC_WIDE_EPILOGUE_WAVE  // (calibration builds: a fixed wave, e.g. 0 = rounds 1 - 2)
  return (void)nb, CUSMC_WIDE_EPILOGUE_WAVE;
#endif
  const WideMap m = wide_map(nb);
  int best = 0, best_load = 1 << 30;
  for (int s = 0; s < 4; ++s) {
    int load = 0;
    for (int w = s; w < 8; w += 4) load += (m.lo[w] + 1) + (m.hi[w] + 1);
    if (load <= best_load) { best_load = load; best = s; }  // (ties: the last SIMD)
  }
  return best;
}
__host__ __device__ constexpr int wide_lo(int nb, int w) { return wide_map(nb).lo[w]; }
__host__ __device__ constexpr int wide_hi(int nb, int w) { return wide_map(nb).hi[w]; }
__host__ __device__ constexpr int wide_pi(int s, int h) { return 2 * h + (s & 1) + 8 * (s >> 1); }

// Block count the kernel runs d with: ceil(d / 16) (12 .. 16 in the library; the map below serves 9 .. 16).  (d <= 176 belongs to the
// tile kernel: 297 us against 382 us here for 1e6 particles -- at 8 blocks the split over waves
// leaves each wave too little work per k-block to cover its fragment loads.)
int mfma_wide_nb(int d) { return (d + 15) / 16; }

// d = 144, 160, ..., 256 with 16-byte aligned rows run unpadded; every other d in (128, 256] and every
// other alignment runs the padded variant (PAD: zero-padded factor, columns >= d zeroed in the LDS staging
// buffer by the loader wave).
bool mfma_wide_supported(int d, const void *X, int64_t ldx)
{
  (void)X;
  // a group of <= 64 rows is addressed through one 32-bit buffer descriptor
  return d > kTileKernelMaxDim && d <= 256 && ldx < (1L << 21);
}
static bool wide_needs_pad(int d, const void *X, int64_t ldx)
{
  return d != 16 * mfma_wide_nb(d) || (uintptr_t)X % 16 != 0 || ldx % 2 != 0;
}

// fragments in the stream of wave w (kernel loop order: kb, s, live members)
static long wide_stream_frags(int nb, int w)
{
  const int lo = wide_lo(nb, w), hi = wide_hi(nb, w);
  if (hi < 0) return 0;  // (a wave without a block: NB < 8 never happens, but the map allows it)
  return lo < 0 ? 4L * (hi + 1) : 4L * (2 * (lo + 1) + (hi - lo));  // both up to kb = lo, then the high block alone
}

static size_t wide_lds_bytes(int nb)
{
  const int tiles = wide_gp(nb) / 16;
  return (size_t)(2 * nb * 4 * tiles * 64 + 2 * wide_waves(nb) * wide_gp(nb) + 32 * nb) * sizeof(double);
}

size_t mfma_wide_frag_doubles(int nb)
{
  size_t n = 0;
  for (int w = 0; w < wide_waves(nb); ++w) n += (size_t)wide_stream_frags(nb, w) * 64;
  return n + 9 * 64;  // zero tail: the kernel prefetches one k-block past a stream's end
}

// streams back to back; a fragment holds, for lane l = (j, h), M[16*cb + j][16*kb + pi(s,h)]
void mfma_wide_pack_frags(const double *M, int d, double *frags)
{
  const int nb = d / 16;
  const size_t total = mfma_wide_frag_doubles(nb);
  for (size_t i = total - 9 * 64; i < total; ++i) frags[i] = 0.0;
  size_t f = 0;
  for (int w = 0; w < wide_waves(nb); ++w) {
    const int lo = wide_lo(nb, w), hi = wide_hi(nb, w);
    for (int kb = 0; kb < nb; ++kb)
      for (int s = 0; s < 4; ++s)
        for (int m = 0; m < 2; ++m) {
          const int cb = m ? hi : lo;
          if (cb < kb) continue;  // (lower triangular; also skips an absent member, cb = -1)
          for (int l = 0; l < 64; ++l) {
            const int j = l & 15, h = l >> 4;
            frags[f * 64 + l] = M[(size_t)(16 * cb + j) * d + 16 * kb + wide_pi(s, h)];
          }
          ++f;
        }
  }
}

static __device__ __forceinline__ double finish_wide(double q, const Epilogue &ep)
{
  double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p_nonneg(q * ep.inv_nu)
                                     : ep.lognorm - 0.5 * q;
  return ep.out_density ? exp(lp) : lp;
}

struct WideStreams { int byte_off[8]; };  // start of each wave's fragment stream

// Calibration hook (scripts/calib/lw_phases.hip; off in the library): every compute wave adds the s_memtime ticks it
// spends per group (0) in the k loop, (1) in the reduction and its LDS write, (2) at the barrier, (3) after it, to
// g_lw_phases[(blockIdx 8 + wave) 4 + phase].
#ifdef CUSMC_LW_PHASES
__device__ unsigned long long g_lw_phases[1024 * 8 * 4];
#define LW_STAMP(k)                                                                                 \
  do {                                                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();                                   \
    if (lane == 0) g_lw_phases[((size_t)blockIdx.x * 8 + w) * 4 + (k)] += now_ - lw_t_;            \
    lw_t_ = now_;                                                                                   \
  } while (0)
#else
#define LW_STAMP(k) do { } while (0)
#endif

// ABL (scripts/calib only; 0 in the library): 1 = fragments not re-fetched, 2 = next group's rows
// not fetched, 3 = neither.  Attribution of stall time; results are wrong by construction.
template <int NB, bool CENTRED, bool SHIFT, int ABL = 0, bool PAD = false>
__global__ __launch_bounds__(64 * (wide_waves(NB) + 1)) void logpdf_mfma_wide_kernel(
    const double *__restrict__ X, long N, long ldx, const double *__restrict__ frags,
    WideStreams streams, long frag_bytes, const double *__restrict__ shift,
    const double *__restrict__ bias, Epilogue ep, double *__restrict__ out, long num_groups,
    int d_true = 16 * NB)
{
  constexpr int WAVES = wide_waves(NB);  // compute waves; wave WAVES is the loader
  constexpr int THREADS = 64 * (WAVES + 1);
  constexpr int GP = wide_gp(NB);
  constexpr int TILES = GP / 16;
  constexpr int XBUF = NB * 4 * TILES * 64;  // doubles per staging buffer
  extern __shared__ double lds[];
  double *sX = lds;                  // [2][NB][2 halves][TILES][64 lanes][2]
  double *sPartial = sX + 2 * XBUF;  // [2][WAVES][GP particles]
  double *sShift = sPartial + 2 * WAVES * GP;
  double *sBias = sShift + 16 * NB;

  for (int i = threadIdx.x; i < 16 * NB; i += THREADS) {
    sShift[i] = shift[i];
    sBias[i] = bias[i];
  }

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long G = gridDim.x;

  if (w == WAVES) {
    // ---------------- loader wave: HBM -> LDS by LDS-DMA, one group ahead ---------------------
    // It shares a SIMD with two compute waves and anything it pushes through the vector ALU or
    // the VGPR file holds up their MFMAs (VGPR staging + ds_write cost 25 % of the kernel), so it
    // moves the rows with `buffer_load_dwordx4 ... lds`: no VGPR data, no VALU.  One instruction
    // deposits a 1 KB slab = (k-block kb, half h2, tile t): lane (p, h) fetches the 16 bytes
    // x[16 t + p][16 kb + 8 h2 + 2h .. +1] -- the operands of k-steps 2 h2, 2 h2 + 1 of that lane --
    // and the DMA puts lane l's chunk at slab + 16 l.  Global side: voffset = (p ldx + 2h) 8
    // (loop-invariant), soffset = (16 t ldx + 16 kb + 8 h2) 8 (scalar).  The compute waves read
    // a slab back with one conflict-free ds_read_b128 per lane.
    const int d_bytes = PAD ? 8 * d_true : 128 * NB;  // (PAD: what lies past the last row's end reads as zero)
    const int voff = (int)(((long)(lane & 15) * ldx + 2 * (lane >> 4)) * 8);
    const int tile_bytes = (int)(16 * ldx * 8);
    auto stage_group = [&](long g, double *buf) {
      long rows = N - g * GP;
      rows = rows < GP ? rows : GP;
      if (rows <= 0) return;
      // the descriptor covers exactly the group's valid bytes: rows past N are out of range
      // (never written or zero: either way only those rows' own, unstored, outputs see them)
      const long bytes = ((rows - 1) * ldx) * 8 + d_bytes;
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(X + g * GP * ldx), 0, (int)bytes, 0x00020000);
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            auto *dst = (__attribute__((address_space(3))) void *)(buf + ((kb * 2 + h2) * TILES + t) * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, t * tile_bytes + 128 * kb + 64 * h2, 0, 0);  // (cache-policy bits nt / sc0 / sc1: no effect, profiles/r03_experiments.txt)
            // Pace the stream: issued as one burst, the group's 64 KB of HBM reads sit in the CU's
            // vector-memory path in front of the compute waves' fragment loads, which then see
            // HBM latency instead of L2 latency (full kernel 672 -> 642 us in the calibration run
            // with ~512 idle cycles after each k-block's slabs; flat between 256 and 900).
            if (t == TILES - 1 && h2 == 1) __builtin_amdgcn_s_sleep(8);
          }
    };
    // PAD, d not a multiple of 16: the last k-block's columns >= d hold the next row's leading values (or
    // whatever lies past the end of X): not this particle's, possibly not finite, so they must not reach
    // the matrix cores even against a zero factor column.  The loader -- idle until the barrier anyway --
    // waits for its DMA and overwrites them with zeros in the staging buffer; the compute waves then run
    // the same instruction stream as the unpadded kernel (masking the operands there, behind a uniform
    // branch per k-step, cost 16 - 21 % of the launch: the branch took the compiler's load scheduling and
    // counted waits across the k-steps with it).
    auto zero_tail = [&](double *buf) {
      if constexpr (PAD) {
        if (d_true == 16 * NB) return;  // (padded for alignment only)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int c0 = 16 * (NB - 1) + 2 * (lane >> 4);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            double *chunk = buf + (((NB - 1) * 2 + h2) * TILES + t) * 128 + 2 * lane;
            if (c0 + 8 * h2 >= d_true) chunk[0] = 0.0;
            if (c0 + 8 * h2 + 1 >= d_true) chunk[1] = 0.0;
          }
      }
    };
    stage_group(blockIdx.x, sX);
    zero_tail(sX);
    __syncthreads();
    int parity = 0;
    for (long g = blockIdx.x; g < num_groups; g += G, parity ^= 1) {
      if (ABL != 2 && ABL != 3) {
        stage_group(g + G, sX + (parity ^ 1) * XBUF);
        zero_tail(sX + (parity ^ 1) * XBUF);
      }
      __syncthreads();
    }
    return;
  }

  // ---------------- compute waves ---------------------------------------------------------------
  const int p = lane & 15, h = lane >> 4;
  // scalar: this wave's output blocks (lo = -1: a single block, hi)
  constexpr WideMap map = wide_map(NB);
  const int lo = map.lo[w], hi = map.hi[w];  // (w is wave-uniform: scalar loads from a constant table)
  const bool single = lo < 0;
  constexpr int tile0 = 0;
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(frags), 0, (int)frag_bytes, 0x00020000);
  const int wbyte0 = streams.byte_off[w];
  const int wlane = lane * 8;
  auto load_w = [&](int fi) -> double {  // fragment fi of this wave's stream
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, wlane, wbyte0 + fi * 512, 0));
  };

  // The fragments of this wave's FIRST k-block are the same for every group: they stay in registers
  // (16 VGPRs of the ~30 the kernel has to spare), so a group starts multiplying as soon as its rows
  // are staged instead of waiting one L2 round trip (~4 % of a group) for them.
  double w0[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) w0[s][i] = single ? load_w(s) : load_w(s * 2 + i);  // (single: one fragment per k-step)

  __syncthreads();  // the first group is staged
  auto run = [&](auto tpw_tag) {
  constexpr int TPW = decltype(tpw_tag)::value;  // tiles per wave = tiles per group: 2, or 3 at NB = 12
  int parity = 0;
#ifdef CUSMC_LW_PHASES
  unsigned long long lw_t_ = __builtin_readcyclecounter();
#endif
  for (long g = blockIdx.x; g < num_groups; g += G, parity ^= 1) {
    LW_STAMP(3);
    // this wave's two tiles of the current buffer: slab (kb, h2, t) holds, per lane, the operands
    // of k-steps 2 h2 and 2 h2 + 1
    const v2d *xw = reinterpret_cast<const v2d *>(sX + parity * XBUF) + tile0 * 64 + lane;

    v4d acc[TPW][2];  // [tile][member: 0 = lo block, 1 = hi block]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      v4d init = v4d{0.0, 0.0, 0.0, 0.0};
      if (!CENTRED && (m ? hi : lo) >= 0) {  // C rows are output dims h + 4r of block cb
        const double *b = sBias + 16 * (m ? hi : lo) + h;
        init = v4d{b[0], b[4], b[8], b[12]};
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t][m] = init;
    }

    // Lookahead.  A k-block is only 8 (two live blocks) or 4 (one) MFMAs per wave, 520 / 260
    // cycles, so operands are requested a whole k-block ahead: wc/wn hold the fragments of the
    // current / next k-block (L2, ~600+ cycles), xa/xb its particle operands (LDS, ~150).  The
    // k loop is unrolled by two so that the sets swap roles without copies.
    double wc[4][2], wn[4][2];
    v2d xa[2][TPW], xb[2][TPW];  // [half][tile]
    int f = 0;  // fragment cursor in this wave's stream (scalar)
    auto load_x = [&](int kb, v2d(&x)[2][TPW]) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const v2d *src = xw + ((kb * 2 + h2) * TILES) * 64;
#pragma unroll
        for (int t = 0; t < TPW; ++t) x[h2][t] = src[t * 64];
      }
    };
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) wc[s][i] = w0[s][i];
    load_x(0, xa);

    // one k-block with M live members (slot i = member 2 - M + i): prefetch the next k-block's
    // operands into (wnext, xnext), then 4 k-steps of 2*M MFMAs from (wcur, xcur)
    auto kblock = [&](auto mtag, int kb, int mn, double(&wcur)[4][2], double(&wnext)[4][2],
                      v2d(&xcur)[2][TPW], v2d(&xnext)[2][TPW]) {
      constexpr int M = decltype(mtag)::value;
      constexpr int C0 = 2 - M;
      f += 4 * M;
      if (ABL != 1 && ABL != 3) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < M; ++i) wnext[s][i] = load_w(f + s * mn + i);
      }
      load_x(kb + 1, xnext);  // one k-block past the end stays inside LDS and is never used
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        double r[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) r[t] = xcur[s >> 1][t][s & 1];
        if (CENTRED && SHIFT) {
          const double sh = sShift[16 * kb + wide_pi(s, h)];
#pragma unroll
          for (int t = 0; t < TPW; ++t) r[t] -= sh;
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            acc[t][C0 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(wcur[s][i], r[t], acc[t][C0 + i], 0, 0, 0);
      }
    };
    // a phase = k-blocks [kb_lo, kb_hi) with M live members; `odd` tells which register set is
    // current on entry, and the phase returns the parity for the next one
    auto run_phase = [&](auto mtag, int kb_lo, int kb_hi, bool odd) -> bool {
      constexpr int M = decltype(mtag)::value;
      int kb = kb_lo;
      if (odd && kb < kb_hi) {  // re-align: one k-block from the (n) sets
        kblock(mtag, kb, (kb + 1 < kb_hi) ? M : 1, wn, wc, xb, xa);
        ++kb;
        odd = false;
      }
#pragma unroll 1
      for (; kb + 1 < kb_hi; kb += 2) {
        kblock(mtag, kb, M, wc, wn, xa, xb);
        kblock(mtag, kb + 1, (kb + 2 < kb_hi) ? M : 1, wn, wc, xb, xa);
      }
      if (kb < kb_hi) {
        kblock(mtag, kb, 1, wc, wn, xa, xb);
        odd = true;
      }
      return odd;
    };
    // (Tried, r02: a wave priority that falls with its progress, so that the two waves of a SIMD finish together
    // instead of the older one in 70 % of a group and the younger one in 92 % -- 1.5 .. 5 % SLOWER; and the extreme
    // pairs on the younger waves -- 1 .. 2 % slower: profiles/r02_experiments.txt.)
    // (Tried: in the single-member phase the 16 fragment registers can hold a ring of four k-blocks,
    // i.e. fragments requested three k-blocks ahead at no register cost.  No gain -- 661..670 us
    // against 654..660 in the calibration run -- so the fragment latency is not what the two streams
    // cost each other.)
    {  // (a single-block wave has no two-member phase: lo + 1 = 0)
      const bool odd = run_phase(std::integral_constant<int, 2>{}, 0, lo + 1, false);
      run_phase(std::integral_constant<int, 1>{}, lo + 1, hi + 1, odd);
    }

    LW_STAMP(0);
    // partial sums of squares over this wave's output blocks, per particle: slot [wave][tile][p]
    // (an absent member's accumulator is zero)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      double qq = 0.0;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) qq = fma(acc[t][m][r], acc[t][m][r], qq);
      qq += __shfl_xor(qq, 16);
      qq += __shfl_xor(qq, 32);
      if (h == 0) sPartial[(parity * WAVES + w) * GP + (tile0 + t) * 16 + p] = qq;
    }
    LW_STAMP(1);
    __syncthreads();  // partials visible; the loader has completed the other buffer
    LW_STAMP(2);
    if (w == wide_epilogue_wave(NB) && lane < GP) {  // fixed summation order over the waves -> bitwise reproducible
      const double *sp = sPartial + parity * WAVES * GP + lane;
      double tot = sp[0];
#pragma unroll
      for (int k = 1; k < WAVES; ++k) tot += sp[k * GP];
      const long row = g * GP + lane;
      if (row < N) out[row] = finish_wide(tot, ep);
    }
  }
  };
  run(std::integral_constant<int, TILES>{});
}

template <int NB>
static WideStreams wide_streams()
{
  WideStreams st{};
  long off = 0;
  for (int w = 0; w < wide_waves(NB); ++w) {
    st.byte_off[w] = (int)(off * 8);
    off += wide_stream_frags(NB, w) * 64;
  }
  return st;
}

template <int NB, bool CENTRED, bool SHIFT, bool PAD>
static hipError_t launch_wide(const double *X, int64_t N, int64_t ldx, int d, const double *frags,
                              const double *shift, const double *bias, const Epilogue &ep,
                              double *out, int num_cus, hipStream_t stream)
{
  auto kern = logpdf_mfma_wide_kernel<NB, CENTRED, SHIFT, 0, PAD>;
  const size_t lds_bytes = wide_lds_bytes(NB);
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  constexpr int GP = wide_gp(NB);
  const long num_groups = (N + GP - 1) / GP;
  long blocks = num_cus;  // one workgroup per CU (two staging buffers fill the LDS)
  if (blocks > num_groups) blocks = num_groups;
  const long frag_bytes = (long)mfma_wide_frag_doubles(NB) * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * (wide_waves(NB) + 1)), lds_bytes, stream, X, (long)N,
                     (long)ldx, frags, wide_streams<NB>(), frag_bytes, shift, bias, ep, out, num_groups, d);
  return hipGetLastError();
}

// calibration entry (scripts/calib/ablate_wide.cpp): d = 256, triangular, no shift, variant abl
hipError_t launch_wide_ablate(int abl, const double *X, int64_t N, const double *frags, const double *zeros,
                              double *out, int blocks, hipStream_t stream)
{
  Epilogue ep{-10.0, 0, 0, 0, 0};
  const long num_groups = (N + 31) / 32;
  const long fb = (long)mfma_wide_frag_doubles(16) * 8;
  const size_t lds = wide_lds_bytes(16);
#define CUSMC_ABL(a)                                                                                           \
  case a:                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(logpdf_mfma_wide_kernel<16, true, false, a>),     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    hipLaunchKernelGGL((logpdf_mfma_wide_kernel<16, true, false, a>), dim3(blocks), dim3(576), lds, stream, X, \
                       (long)N, 256L, frags, wide_streams<16>(), fb, zeros, zeros, ep, out, num_groups);  \
    break;
  switch (abl) { CUSMC_ABL(0) CUSMC_ABL(1) CUSMC_ABL(2) CUSMC_ABL(3) }
#undef CUSMC_ABL
  return hipGetLastError();
}

int wide_occupancy_probe()
{
  int n = -1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(logpdf_mfma_wide_kernel<16, true, false, 0>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes(16));
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, logpdf_mfma_wide_kernel<16, true, false, 0>, 576,
                                                     wide_lds_bytes(16));
  return n;
}

hipError_t launch_logpdf_mfma_wide(const double *X, int64_t N, int64_t ldx, int d, bool centred,
                                   bool has_shift, const double *frags, const double *shift,
                                   const double *bias, const Epilogue &ep, double *out,
                                   int num_cus, hipStream_t stream)
{
  if (N <= 0) return hipSuccess;
  const bool pad = wide_needs_pad(d, X, ldx);
#define CUSMC_WPAD(nb, t, s)                                                                       \
  (pad ? launch_wide<nb, t, s, true>(X, N, ldx, d, frags, shift, bias, ep, out, num_cus, stream)   \
       : launch_wide<nb, t, s, false>(X, N, ldx, d, frags, shift, bias, ep, out, num_cus, stream))
#define CUSMC_WIDE(nb)                                                                             \
  case nb:                                                                                         \
    if (!centred) return CUSMC_WPAD(nb, false, false);                                              \
    return has_shift ? CUSMC_WPAD(nb, true, true) : CUSMC_WPAD(nb, true, false);
  switch (mfma_wide_nb(d)) {
#if CUSMC_TILE_MAX_NB < 11  // (calibration builds only: the tile kernel takes these in the library)
    CUSMC_WIDE(9) CUSMC_WIDE(10) CUSMC_WIDE(11)
#endif
    CUSMC_WIDE(12) CUSMC_WIDE(13) CUSMC_WIDE(14) CUSMC_WIDE(15) CUSMC_WIDE(16)
  }
#undef CUSMC_WIDE
#undef CUSMC_WPAD
  return hipErrorInvalidValue;
}

}  // namespace cusmc
