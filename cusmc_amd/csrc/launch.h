// Host-callable launchers of the gfx950 kernels (one per .hip file under kernels/).
// Internal to libcusmc_hip.so; the public surface is include/cusmc_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

namespace cusmc {

// Raises a kernel's dynamic-LDS limit above the default 64 KB, per (kernel instantiation, device): `cfg` is
// the instantiation's own record of the byte count configured on each device (a context may live on any
// device of the process, and the attribute is per device).  Kernels that size their LDS from d at run
// time come back with a larger request later (d = 130 needs 67 KB, d = 250 128 KB): the attribute is
// raised again whenever the request exceeds what was configured -- a bare "configured" bit would let
// the second launch fail with hipErrorInvalidValue.
struct LdsConfig {
  std::atomic<unsigned> bytes[64] = {};
};
inline hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes, LdsConfig &cfg)
{
  if (bytes <= 64 * 1024) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && cfg.bytes[dev].load(std::memory_order_relaxed) >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && tracked) {
    unsigned cur = cfg.bytes[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !cfg.bytes[dev].compare_exchange_weak(cur, (unsigned)bytes, std::memory_order_relaxed)) {}
  }
  return e;
}


// Epilogue of the quadratic form q = |z|^2:
//   mvn: lognorm - q/2                              (src/statistics.cc.cpp:179, as a log)
//   mvt: lognorm - (nu+d)/2 * log1p(q/nu)           (src/statistics.cc.cpp:308-310, as a log)
struct Epilogue {
  double lognorm;
  double half_nu_plus_d;  // mvt only
  double inv_nu;          // mvt only
  int kind;               // CUSMC_MVN / CUSMC_MVT
  int out_density;        // exponentiate at the end
};

// Every log-pdf kernel computes, per particle x (d doubles):
//     z = bias + M (x - shift),   q = z.z,   out = epilogue(q)
// centred form  (pdf(y,F)):   M = W = L^-1 (lower triangular), shift = F mu, bias = 0
// affine form   (reweight_G): M = -W F (dense),                shift = 0,    bias = W y
//   -- for d >= 16 rotated on the host to M = L (lower triangular), bias = Q^T W y with -W F = Q L:
//   |z|^2 does not change and the matrix-core kernels only ever see triangular factors

// Largest block count of the tile kernel (logpdf_mfma.hip): d <= 176.  Its factor lives in LDS from
// NB = 5 up -- 135 KB at NB = 11, and NB = 12 (160 KB) no longer fits beside the rest; the wide kernel
// (logpdf_mfma_wide.hip) serves 176 < d <= 256.  Where both can run the tile kernel wins (1e6 particles,
// scripts/d128_ab.py: d = 144 380 against 517 us, d = 160 467 / 579, d = 176 547 / 636): a wave that owns
// a whole particle tile reuses every factor fragment it reads from LDS for NB..1 block-products, the wide
// kernel's waves fetch theirs from L2 for two.  (-DCUSMC_TILE_MAX_NB=8 restores the old split for A/B runs.)
#ifndef CUSMC_TILE_MAX_NB
#define CUSMC_TILE_MAX_NB 11
#endif
constexpr int kTileKernelMaxDim = 16 * CUSMC_TILE_MAX_NB;

// --- kernels/logpdf_mfma.hip : d = 16*NB, v_mfma_f64_16x16x4_f64 -------------------------------
// Number of 512-byte B fragments the kernel expects in `frags` (LDS image, kernel loop order).
int mfma_num_frags(int nb, bool tri);
// Is (d, X, ldx) servable by the MFMA kernel?
bool mfma_supported(int d, const void *X, int64_t ldx);
// Host-side packing of M (d x d row-major) into the fragment order.
void mfma_pack_frags(const double *M, int d, bool tri, double *frags);
// frags: the LOWER TRIANGULAR factor packed with tri = true.  centred: shift, no bias; !centred:
// bias, no shift.  has_shift = false promises the shift vector is all zeros.
// pool: tail-pool state for the assembly kernel of the d = 64 headline shape (kernels/logpdf_nb4_gfx950.s), ONE PER
// STREAM: nb4_pool_bytes() of device memory, zeroed once, and the number of launches that used it (consecutive
// launches alternate between its two counter blocks); NULL: the compiled kernel
constexpr int kNb4PoolCounters = 32;
struct Nb4Pool {
  unsigned *dev = nullptr;
  unsigned launches = 0;
};
size_t nb4_pool_bytes();
hipError_t launch_logpdf_mfma(const double *X, int64_t N, int64_t ldx, int d, bool centred,
                              bool has_shift, const double *frags, const double *shift,
                              const double *bias, const Epilogue &ep, double *out, int num_cus,
                              hipStream_t stream, Nb4Pool *pool = nullptr);

// --- kernels/logpdf_mfma_wide.hip : 128 < d <= 256, output blocks split over the waves -----------
bool mfma_wide_supported(int d, const void *X, int64_t ldx);
int mfma_wide_nb(int d);  // 16-column blocks the wide kernel runs d with: ceil(d / 16) = 9 .. 16
size_t mfma_wide_frag_doubles(int nb);
void mfma_wide_pack_frags(const double *M, int d, double *frags);
hipError_t launch_logpdf_mfma_wide(const double *X, int64_t N, int64_t ldx, int d, bool centred,
                                   bool has_shift, const double *frags, const double *shift,
                                   const double *bias, const Epilogue &ep, double *out,
                                   int num_cus, hipStream_t stream);

// --- kernels/logpdf_generic.hip : any d <= 639, lane = particle --------------------------------
bool generic_supported(int d);
hipError_t launch_logpdf_generic(const double *X, int64_t N, int64_t ldx, int d, bool tri,
                                 const double *M, const double *shift, const double *bias,
                                 const Epilogue &ep, double *out, int num_cus,
                                 hipStream_t stream);

// --- kernels/resample.hip ------------------------------------------------------------------------
// the chain over log-weights: accept iff u <= exp(lw[j] - lw[k])
hipError_t launch_metropolis_log(const double *lw, uint32_t N, uint32_t B, uint64_t seed, uint32_t step,
                                 uint32_t first, uint32_t count, uint32_t *a, int num_cus, hipStream_t stream);
bool metropolis_wants_hiwords(uint32_t N);  // weight table too big for one XCD's L2
bool metropolis_wants_lds_table(uint32_t N, uint32_t B, uint32_t count);  // head of the truncated table in LDS (needs whi)
hipError_t launch_hiwords(const double *w, uint32_t N, uint32_t *whi, int num_cus, hipStream_t stream);
// whi: high words of w (launch_hiwords), or NULL to gather from the doubles
hipError_t launch_metropolis(const double *w, const uint32_t *whi, uint32_t N, uint32_t B, uint64_t seed,
                             uint32_t step, uint32_t first, uint32_t count, uint32_t *a, int num_cus,
                             hipStream_t stream);

// --- kernels/propagate.hip -----------------------------------------------------------------------
// X_out[i-first] = [diag(c)] Q (scale xi) + (G ? G X_prev[a ? a[i-first] : i] : m0)
hipError_t launch_propagate(int kind, float nu, const double *X_prev, const uint32_t *a,
                            const double *G, const double *Q, const double *m0, int d,
                            double scale, uint64_t seed, uint32_t step, uint32_t domain,
                            uint32_t first, uint32_t count, double *X_out, int num_cus,
                            hipStream_t stream);

// --- kernels/propagate_mfma.hip : d = 16*NB <= 64 ------------------------------------------------
// fragsQ / fragsG: dense mfma_pack_frags images of Q and G (fragsG == nullptr: initial draw, + m0);
// g_is_diagonal: fragsG is the d-vector diag(G) instead (no second product)
bool propagate_mfma_supported(int d, const void *X_prev, const void *X_out);
hipError_t launch_propagate_mfma(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *fragsQ, const double *fragsG, bool g_is_diagonal, const double *m0,
                                 int d, double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream);
// the same with a LOWER TRIANGULAR Q, fragsQ packed with tri = true (kernels/propagate_mfma_tri.hip)
hipError_t launch_propagate_mfma_tri(int kind, float nu, const double *X_prev, const uint32_t *a,
                                     const double *fragsQ, const double *fragsG, bool g_is_diagonal, const double *m0,
                                     int d, double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                     uint32_t first, uint32_t count, double *X_out, int num_cus,
                                     hipStream_t stream);

// --- kernels/propagate_mfma_wide.hip : 128 < d <= 256, output blocks split over the waves -----------------
// fragsQ: dense mfma_pack_frags image of Q (zero-padded to 16*ceil(d/16)); tail: the same of G (mode 1),
// diag(G) (mode 4) or m0 (mode 0)
bool propagate_mfma_wide_supported(int d, const void *X_prev, const void *X_out);
hipError_t launch_propagate_mfma_wide(int kind, float nu, const double *X_prev, const uint32_t *a, const double *fragsQ,
                                      const double *tail, int mode, int d, double scale, uint64_t seed, uint32_t step,
                                      uint32_t domain, uint32_t first, uint32_t count, double *X_out, int num_cus,
                                      hipStream_t stream);
// the same with a LOWER TRIANGULAR Q, fragsQ packed with tri = true (kernels/propagate_mfma_wide_tri.hip)
hipError_t launch_propagate_mfma_wide_tri(int kind, float nu, const double *X_prev, const uint32_t *a, const double *fragsQ,
                                          const double *tail, int mode, int d, double scale, uint64_t seed, uint32_t step,
                                          uint32_t domain, uint32_t first, uint32_t count, double *X_out, int num_cus,
                                          hipStream_t stream);

// diagonal G (or none: m0) and diagonal Q, any d: one lane per component pair
hipError_t launch_propagate_diag(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *gdiag, const double *qdiag, const double *m0, int d,
                                 double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream);

// dense G / Q, 128 < d: one workgroup per particle; GT / QT are the transposed factors
hipError_t launch_propagate_rows(int kind, float nu, const double *X_prev, const uint32_t *a,
                                 const double *GT, const double *QT, const double *m0, int d,
                                 double scale, uint64_t seed, uint32_t step, uint32_t domain,
                                 uint32_t first, uint32_t count, double *X_out, int num_cus,
                                 hipStream_t stream);

// --- a particle array sharded contiguously over devices (cusmc_pf_run_multi_host) -------------------------------
constexpr int kMaxShards = 16;
struct ShardTable {
  const double *base[kMaxShards];    // shard r's rows of x_{t-1} (a pointer valid on the launching device)
  uint32_t first[kMaxShards + 1];    // shard r owns particles [first[r], first[r+1])
  int n;
};
// the sharded form of the fused step: rows of x_{t-1} read from their owners, w_t (and its high words, the
// resampler's gather table of the NEXT step) written into every shard's copy of the weight vector
struct ShardStep {
  ShardTable x;                      // x.n == 0: not sharded
  double *w_dst[kMaxShards];         // shard r's full-length w_t            (entry `first + t` written)
  uint32_t *whi_dst[kMaxShards];     // shard r's full-length high words of w_t
};

// --- kernels/pf_step.hip : resample + propagate + reweight in one launch, d <= 8 -----------------
bool pf_step_supported(int d);
// sharded != NULL: X_prev is ignored (rows come through sharded->x) and the weights also go to sharded->w_dst / whi_dst
hipError_t launch_pf_step(int kind, float nu, const double *w_prev, const uint32_t *w_prev_hi,
                          const double *X_prev,
                          uint32_t N, int d, uint32_t B, const double *G, const double *Q,
                          double scale, bool tri, const double *M, const double *shift,
                          const double *bias, const Epilogue &ep, uint64_t seed, uint32_t step,
                          uint32_t first, uint32_t count, uint32_t *a_out, double *X_out,
                          double *w_out, int num_cus, hipStream_t stream, const ShardStep *sharded = nullptr);

// --- kernels/gather.hip : ancestor rows of a particle array sharded over devices -----------------------
// out[i][:] = x[a[i]][:] for i < count, row a[i] read from the shard that owns it
hipError_t launch_gather_rows_sharded(const ShardTable &tab, const uint32_t *a, uint32_t count, int d, double *out,
                                      int num_cus, hipStream_t stream);

// --- kernels/percov.hip : per-particle covariances, d <= 16 (lane = matrix, triangle in LDS) -------
bool percov_supported(int d);
hipError_t launch_cholesky_batched(const double *S, int64_t N, int d, double *L, double *logdet, int *info,
                                   int num_cus, hipStream_t stream);
// ep.lognorm: the normalising constant without the determinant term (-logdet_i / 2 is added per particle)
hipError_t launch_logpdf_percov(const double *X, int64_t N, int64_t ldx, const double *mu, int64_t ldmu,
                                const double *S, int d, const Epilogue &ep, double *out, int *info, int num_cus,
                                hipStream_t stream);

}  // namespace cusmc
