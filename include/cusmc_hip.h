/*
 * cusmc_hip.h -- C ABI of libcusmc_hip.so: the MI355X (gfx950) implementation of CuSMC's
 * per-particle likelihood / proposal / accept-reject hot path.
 *
 * This is the lower drop-in boundary.  The reference has no C ABI of its
 * own; its nearest analogue is the five C++ device-wrapper prototypes in
 * inst/include/distributions/mvn_dist.hpp:19-54 (Eigen arguments, void return, death via
 * FATAL/Rcpp::stop, inst/include/support.cuh:9-32).  Each entry point below names the
 * reference interface it stands in for.  The Rcpp glue that binds these to the six registered
 * R symbols (src/RcppExports.cpp:105-113) is in rcpp/src; see INTEGRATION.md.
 *
 * Conventions
 *   - Every function returns an int status: CUSMC_OK (0) or a CUSMC_E* code; the text of the
 *     last failure on the calling thread is cusmc_last_error().  No exception crosses the ABI.
 *   - Plain pointers and sizes only.  `_dev` arguments are device pointers on the context's
 *     GPU; everything else is host memory owned by the caller.  Opaque handles own device
 *     memory and must be destroyed by the caller.
 *   - All arithmetic is fp64.  Matrices are dense ROW-major, M[i*d+j] (an Eigen column-major
 *     matrix is passed as its transpose: see rcpp/src/glue.hpp).  Particle batches are N x d
 *     row-major with leading dimension ldx >= d: each particle's d doubles contiguous -- for an
 *     R `d x N` matrix (columns = particles, src/run.rcpp.cpp:91) that is the buffer as it is.
 *   - nu is a C float, as in the reference (inst/include/statistics.hpp:30,199).
 *   - Work is enqueued on the context's HIP stream; `_dev` calls return without waiting
 *     (use cusmc_ctx_synchronize), `_host` calls return when the output buffer is filled.
 *   - One context per host thread (thread-compatible, not thread-safe) -- all R needs.
 *   - Destroying a context orphans the distributions created on it (their device buffers go with
 *     it): a later call through such a handle returns CUSMC_EINVAL, and cusmc_dist_destroy still
 *     accepts it -- finalizers (R's at session end, Python's at interpreter exit) run in any order.
 *   - RNG: counter-based Philox4x32-10, key = seed, counter = (index, sub, step, domain); the
 *     full contract is in DESIGN.md section "RNG contract" and restated in oracle/cusmc_oracle.c.
 */
#ifndef CUSMC_HIP_H
#define CUSMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUSMC_OK 0
#define CUSMC_EINVAL 1    /* bad argument (NULL, d <= 0, ldx < d, unknown option string ...) */
#define CUSMC_ENOTSPD 2   /* covariance / scale matrix is not symmetric positive definite   */
#define CUSMC_EHIP 3      /* a HIP runtime call failed (text in cusmc_last_error)           */
#define CUSMC_ENODEVICE 4 /* no usable gfx950 device                                          */
#define CUSMC_ERANGE 5    /* size beyond what the kernels support (d > CUSMC_MAX_DIM ...)    */

/* Largest state dimension.  d <= 256 runs on the tuned kernels; 256 < d <= 384 is served by the
 * shape-agnostic ones (lane-per-particle log-density, workgroup-per-particle proposal draws): correct, an order of
 * magnitude slower per particle.  The reference has no limit of its own below the overflow of tgamma() in its Student-t
 * norm at d ~ 340 (src/statistics.cc.cpp:302, 332-340); log-densities here go through lgamma and do not overflow. */
#define CUSMC_MAX_DIM 384

/* distribution kinds -- the keys of the reference's Distributions registry
 * ("mvn", "mvt": src/mcmc.cpp:53-58) */
#define CUSMC_MVN 0
#define CUSMC_MVT 1

/* output flags for the density entry points */
#define CUSMC_OUT_LOG 0     /* log-density (the build's internal quantity)                  */
#define CUSMC_OUT_DENSITY 1 /* density, what the reference returns (statistics.cc.cpp:179) */

typedef struct cusmc_ctx cusmc_ctx;   /* device + stream + scratch                            */
typedef struct cusmc_dist cusmc_dist; /* a StatisticalDistribution instance (mu, Sigma[, nu]) */

/* ---- library / context ---------------------------------------------------------------- */

const char *cusmc_version(void);
/* Version of the counter-based RNG contract the draws follow: which Philox block and
 * which words of it feed which draw.  A seed reproduces a result only under the same contract version.
 * 2 (round 3): chi-square draws keyed by the component pair -- closed form for nu = 2, 4, pair-shared
 * Marsaglia-Tsang attempts otherwise.  3 (round 3): the resampler takes ONE Philox block per TWO chain steps (32
 * leading bits of u and a 32-bit index candidate per step, both completed exactly from further blocks in the rare
 * cases where they cannot decide); everything else as contract 1.  4 (round 3): the closed-form chi-square
 * draw for every integer nu <= 16 (products of nu / 2 uniforms, a squared normal on top for odd nu); nu = 2, 4
 * and every non-integer nu draw what contract 3 drew.  (The reference has no contract: it
 * reseeds from std::random_device per call, src/statistics.cc.cpp:231-232, 360-361.) */
int cusmc_rng_contract(void);
const char *cusmc_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int cusmc_device_count(void);

/* Key of the `call`-th R-level call of a session seeded with `seed` (SplitMix64 output call + 1).  The
 * reference's R-level MVN() / MVT() / metropolis_hastings() / run() take no seed and reseed from
 * std::random_device on every call (src/samplers.cpp:10-11, src/statistics.cc.cpp:231-232,361-362);
 * the host layers keep (seed, call counter) and pass this key as the `seed` of the entry points
 * below, so that no two calls of a session share Philox counters.  Pure host function; returns the
 * key, not a status. */
uint64_t cusmc_stream_key(uint64_t seed, uint64_t call);

/* device < 0: keep the calling thread's current device.  The reference has no context: it
 * uses device 0 / the default stream implicitly and cudaDeviceReset()s after each call
 * (src/mvn_dist.cu.cpp:788). */
int cusmc_ctx_create(int device, cusmc_ctx **out);
int cusmc_ctx_destroy(cusmc_ctx *ctx);
/* Adopt a caller-owned hipStream_t (NULL = the default stream). */
int cusmc_ctx_set_stream(cusmc_ctx *ctx, void *hip_stream);
int cusmc_ctx_synchronize(cusmc_ctx *ctx);
/* multiProcessorCount of the context's device. */
int cusmc_ctx_num_cus(cusmc_ctx *ctx, int *out);

/* ---- distributions: MultiVariateNormalDistribution / MultiVariateTStudentDistribution ---
 * getInstance(params) -- src/statistics.cc.cpp:164-168, :288-292.  Factors Sigma once
 * (Cholesky, W = L^-1, log det) and uploads the factor in the kernels' fragment order.
 * mu may be NULL (zero mean, as reweight_G sets it: src/mcmc.cpp:188).  nu ignored for MVN. */
int cusmc_dist_create(cusmc_ctx *ctx, int kind, const double *mu, const double *sigma, int d,
                      float nu, cusmc_dist **out);
int cusmc_dist_destroy(cusmc_dist *dist);

/* getNorm() -- src/statistics.cc.cpp:205-211 (mvn), :332-340 (mvt); returned as a log. */
int cusmc_dist_lognorm(const cusmc_dist *dist, double *out);
int cusmc_dist_logdet(const cusmc_dist *dist, double *out);

/* Batched pdf(y, F) -- src/statistics.cc.cpp:183-196 (mvn), :295-311 (mvt):
 *     out[i] = log p( X[i,:] ;  F mu, Sigma )            r_i = x_i - F mu
 * F == NULL means the identity (what MVNPDF()/MVTPDF() pass: src/mvn_dist.rcpp.cpp:55,
 * src/mvt_dist.rcpp.cpp:64). */
int cusmc_dist_pdf_dev(cusmc_dist *dist, const double *X_dev, int64_t N, int64_t ldx,
                       const double *F, int flags, double *out_dev);
int cusmc_dist_pdf_host(cusmc_dist *dist, const double *X, int64_t N, int64_t ldx,
                        const double *F, int flags, double *out);

/* reweight_G -- src/mcmc.cpp:162-237 (the pdf(y) overload, :171-180 / :313-324, applied to
 * y_t - F x_i; replaces mvn_pdf_kernel_wrapper / mvt_pdf_kernel_wrapper,
 * inst/include/distributions/mvn_dist.hpp:19-27,38-46):
 *     out[i] = log p( y - F X[i,:] ;  0, Sigma )
 * The distribution's own mu is not used (the reference sets it to zero there). */
int cusmc_dist_reweight_dev(cusmc_dist *dist, const double *X_dev, int64_t N, int64_t ldx,
                            const double *y, const double *F, int flags, double *out_dev);
int cusmc_dist_reweight_host(cusmc_dist *dist, const double *X, int64_t N, int64_t ldx,
                             const double *y, const double *F, int flags, double *out);

/* The two host-pointer density calls with the ROWS sharded contiguously over `ndev` GPUs of one node: one host
 * thread, one library-owned context and one replica of `dist` (factored and uploaded once, kept with the handle)
 * per shard, every device fed over its own PCIe link, results written straight into `out`.  Each row's value
 * is the single-device value bit for bit.  Fewer shards are used when N is small (about 1000 rows per shard at
 * least); a device may be listed more than once.  cusmc_dist_pdf_host / cusmc_dist_reweight_host take this route
 * themselves when the environment holds CUSMC_DEVICES="0,1,..." -- so the R-level MVNPDF() / MVTPDF() on a d x N
 * matrix (src/mvn_dist.rcpp.cpp:52-58, src/mvt_dist.rcpp.cpp:60-66) use the node without a new argument.  The
 * reference has no multi-device code. */
int cusmc_dist_pdf_multi_host(cusmc_dist *dist, const int *devices, int ndev, const double *X, int64_t N,
                              int64_t ldx, const double *F, int flags, double *out);
int cusmc_dist_reweight_multi_host(cusmc_dist *dist, const int *devices, int ndev, const double *X, int64_t N,
                                   int64_t ldx, const double *y, const double *F, int flags, double *out);

/* ---- Metropolis resampler: Sampler::metropolis_hastings -- src/samplers.cpp:7-36 ----------
 * For each i in [first, first+count):  k = i; repeat B times { u ~ U[0,1); j ~ UnifInt[0,N);
 * if (u <= w[j] / w[k]) k = j; }  a[i-first] = k.      (0-based ancestors, as the reference.)
 * w has N entries (the full weight vector); [first, first+count) is this caller's shard of
 * the output (first = 0, count = N on one GPU).  `step` is the reference's t. */
int cusmc_metropolis_dev(cusmc_ctx *ctx, const double *w_dev, uint32_t N, uint32_t B,
                         uint64_t seed, uint32_t step, uint32_t first, uint32_t count,
                         uint32_t *a_dev);
int cusmc_metropolis_host(cusmc_ctx *ctx, const double *w, uint32_t N, uint32_t B, uint64_t seed,
                          uint32_t step, uint32_t *a);

/* The same chain over LOG-weights (what the log-density entry points produce with CUSMC_OUT_LOG):
 *     accept  iff  u <= exp(logw[j] - logw[k])
 * -- the reference's test with w = exp(logw), without the underflow that turns densities at large d
 * into 0 / 0.  The reference stores densities (reweight_G, src/mcmc.cpp:208) and has no such entry;
 * this is the lower boundary's extension named in SURVEY.md 8(b).  -inf is a zero weight. */
int cusmc_metropolis_log_dev(cusmc_ctx *ctx, const double *logw_dev, uint32_t N, uint32_t B,
                             uint64_t seed, uint32_t step, uint32_t first, uint32_t count,
                             uint32_t *a_dev);
int cusmc_metropolis_log_host(cusmc_ctx *ctx, const double *logw, uint32_t N, uint32_t B,
                              uint64_t seed, uint32_t step, uint32_t *a);

/* The chains sharded contiguously over `ndev` GPUs (BASELINE configs[3]: "chains sharded over 8 x MI355X"): every
 * device receives the whole weight vector and runs chains [first_r, first_r + count_r) into a + first_r.  The
 * draws are keyed by the global chain index: the ancestors are the single-device ones bit for bit.
 * log_weights != 0: w holds log-weights (cusmc_metropolis_log_*).  cusmc_metropolis_host /
 * cusmc_metropolis_log_host take this route themselves under CUSMC_DEVICES="0,1,..." -- what the R-level
 * metropolis_hastings() (src/samplers.rcpp.cpp:35-55) binds. */
int cusmc_metropolis_multi_host(const int *devices, int ndev, const double *w, uint32_t N, uint32_t B,
                                uint64_t seed, uint32_t step, int log_weights, uint32_t *a);

/* ---- proposal draws ------------------------------------------------------------------------
 * propagate_K -- src/mcmc.cpp:90-160 (replaces mvn_sample_kernel_wrapper /
 * mvt_sample_kernel_wrapper, mvn_dist.hpp:29-32,48-54):
 *     x_t[i] = [diag(c_i)] Q (scale * xi_i) + G x_prev[a[i]],   xi_i ~ N(0, I)
 * c_i,j = sqrt(nu / chi2_nu) per component for kind == CUSMC_MVT (src/statistics.cc.cpp:385-386,
 * 411).  Q is the dense square-root factor the caller supplies (eigenSolver:
 * src/linear_algebra.cpp:10-23, or cusmc_eigen_sqrt below); a Q whose upper triangle is zero (a Cholesky
 * factor: the same law) is recognised and multiplied as a triangle -- half the matrix-core work of Q xi from
 * d = 32 up, same values.  scale = 1 draws from N(mu, QQ^T);
 * scale = sqrt(3) reproduces the distribution of the reference's CPU transform
 * (src/statistics.cc.cpp:245-256).  a_dev == NULL means a[i] = i.
 * Rows [first, first+count) of the output are produced (X_out_dev has `count` rows) and key the
 * draws; X_prev has N rows, named by a[] (values < N) -- or, for a_dev == NULL, by i itself, in
 * which case first + count <= N is required. */
int cusmc_propagate_dev(cusmc_ctx *ctx, int kind, float nu, const double *X_prev_dev,
                        const uint32_t *a_dev, uint32_t N, int d, const double *G,
                        const double *Q, double scale, uint64_t seed, uint32_t step,
                        uint32_t first, uint32_t count, double *X_out_dev);

/* initialize() draws -- src/mcmc.cpp:44-88 (replaces mvn_sample_kernel_wrapper(init),
 * mvn_dist.hpp:34-37):   x_0[i] = [diag(c_i)] Q (scale * xi_i) + m0 */
int cusmc_initialize_dev(cusmc_ctx *ctx, int kind, float nu, const double *m0, const double *Q,
                         int d, double scale, uint64_t seed, uint32_t first, uint32_t count,
                         double *X_out_dev);

/* StatisticalDistribution::sample(draws, Q, n_iter) as the R-level MVN()/MVT() call it
 * (src/mvn_dist.rcpp.cpp:31-37, src/mvt_dist.rcpp.cpp:28-49), batched: `count` draws
 *     X_out[i] = [diag(c_i)] Q (scale * xi_i) + mu,   index = i, counter step = `step`
 * into host memory (count x d row-major). */
int cusmc_sample_host(cusmc_ctx *ctx, int kind, float nu, const double *mu, const double *Q, int d,
                      double scale, uint64_t seed, uint32_t step, uint32_t count, double *X_out);

/* eigenSolver -- src/linear_algebra.cpp:10-23:  Q = V sqrt(Lambda), Q Q^T = sigma.  Host only. */
int cusmc_eigen_sqrt(const double *sigma, int d, double *Q);

/* ---- per-particle covariances (d <= 16) -- SURVEY.md 8(f) row 4 --------------------------------
 * The reference has one covariance per distribution object (distParams_t, inst/include/
 * statistics.hpp:22-34) but factors it inside every pdf() call (Sigma.determinant() and
 * Sigma.inverse(), src/statistics.cc.cpp:176-177, 301, 306); these entry points serve models whose
 * covariance differs from particle to particle.  sigma: N matrices, d x d row-major, back to back
 * (only the lower triangles are read).
 *   cusmc_chol_batched_*:   L[i] = lower Cholesky factor (upper part zeroed), logdet[i] = log det
 *                           Sigma_i, info[i] = 0 or 1 + the index of the first non-positive pivot
 *                           (L[i] and logdet[i] are then not finite).  logdet / info may be NULL.
 *   cusmc_logpdf_percov_*:  out[i] = log p(X[i,:]; mu_i, Sigma_i) -- multivariate Normal, or
 *                           Student-t with nu degrees of freedom (float, as the reference's nu);
 *                           mu: N x ldmu, or ONE shared d-vector when ldmu == 0, or NULL (zero).
 *                           flags as above; NaN where Sigma_i is not positive definite. */
int cusmc_chol_batched_dev(cusmc_ctx *ctx, const double *sigma_dev, int64_t N, int d, double *L_dev,
                           double *logdet_dev, int32_t *info_dev);
int cusmc_chol_batched_host(cusmc_ctx *ctx, const double *sigma, int64_t N, int d, double *L,
                            double *logdet, int32_t *info);
int cusmc_logpdf_percov_dev(cusmc_ctx *ctx, int kind, float nu, const double *X_dev, int64_t N,
                            int64_t ldx, const double *mu_dev, int64_t ldmu,
                            const double *sigma_dev, int d, int flags, double *out_dev,
                            int32_t *info_dev);
int cusmc_logpdf_percov_host(cusmc_ctx *ctx, int kind, float nu, const double *X, int64_t N,
                             int64_t ldx, const double *mu, int64_t ldmu, const double *sigma, int d,
                             int flags, double *out, int32_t *info);

/* ---- one filter time step: the body of MCMC()'s loop -- src/mcmc.cpp:292-308 -----------------
 *     a[i]   = Metropolis chain i over w_prev                 (cusmc_metropolis_dev)
 *     x_t[i] = [diag(c_i)] Q (scale * xi_i) + G x_prev[a[i]]  (cusmc_propagate_dev, kind/nu of the proposal)
 *     w_t[i] = pdf_{0,Sigma_obs}( y - F x_t[i] )              (cusmc_dist_reweight_dev on `obs`)
 * for i in [first, first+count).  w_prev and X_prev hold all N particles of step t-1; the three
 * outputs hold this caller's `count` rows.  For d <= 8 this is ONE launch (the state never leaves
 * registers between the proposal and the weight); otherwise it is the three calls above, with
 * identical results either way.  flags: CUSMC_OUT_LOG or CUSMC_OUT_DENSITY (what reweight_G
 * stores, mcmc.cpp:208). */
int cusmc_pf_step_dev(cusmc_dist *obs, int kind, float nu, const double *w_prev_dev,
                      const double *X_prev_dev, uint32_t N, const double *G, const double *Q,
                      const double *y, const double *F, uint32_t B, double scale, uint64_t seed,
                      uint32_t step, uint32_t first, uint32_t count, uint32_t *a_out_dev,
                      double *X_out_dev, double *w_out_dev, int flags);

/* ---- the filter: particle_filter() -- src/particle_filter.cpp:6-39, MCMC() mcmc.cpp:239-309 -
 * Device-resident time loop: initialize, then for t = 1..T-1: resample(w_{t-1}) -> propagate
 * -> reweight.  Y is T x d (row t = y_t; the reference stores Y.col(t): run.rcpp.cpp:91).
 * Outputs (host, any may be NULL): X T x N x d, w T x N (densities, unnormalised, as
 * run.rcpp.cpp:110-116 returns them), a T x N (row 0 unwritten in the reference; zeros here).
 * resampler must be "metropolis", distribution "mvn" or "mvt" (mcmc.cpp:252-266); anything
 * else is CUSMC_EINVAL (the reference throws bad_function_call).
 * The proposals' square roots of C0 and W are eigenSolver's, as MCMC() computes them (mcmc.cpp:70-71, 280);
 * with CUSMC_PROPOSAL_FACTOR=cholesky in the environment they are the lower Cholesky factors instead (same
 * law, other realisations; the triangular proposal kernels from d = 32 up). */
int cusmc_pf_run_host(cusmc_ctx *ctx, const double *Y, uint32_t N, int d, uint32_t T,
                      const double *m0, const double *C0, const double *F, const double *G,
                      const double *V, const double *W, float df, const char *resampler,
                      const char *distribution, uint32_t B, double scale, uint64_t seed,
                      double *X_out, double *w_out, uint32_t *a_out);

/* The same filter with the particles sharded over `ndev` GPUs of one node (contiguous shards, one host
 * thread + library-owned context per shard, peer access over xGMI) -- the loop of src/mcmc.cpp:292-308 being
 * sharded; the reference has no multi-device code.  Exact algorithm, results bit-identical to one device by
 * construction: every draw is keyed by the global particle index.  Per step a device receives the other shards'
 * weights (8 (N - N/R) bytes) and only the rows x_{t-1}[a_i] its own ancestors name (at most 8 d N/R bytes).
 * The steps are ordered ON THE DEVICES (a stream waits for its peers' "step t-1 published" events); the host
 * threads enqueue the whole loop without blocking on a GPU.  A device may be listed more than once (several
 * shards on one GPU: how a one-GPU box rehearses this path -- and the only way it has been run so far: the
 * branches that differ between distinct devices, hipDeviceEnablePeerAccess, hipMemcpyPeerAsync and the peer
 * reads of kernels/gather.hip, are UNVERIFIED ON HARDWARE until a run on two physical GPUs is recorded;
 * tests/test_gpu_parity.py::test_multi_device_run_on_two_physical_gpus does it where two are visible).
 * cusmc_pf_run_host itself takes this route when the environment holds CUSMC_DEVICES="0,1,...". */
int cusmc_pf_run_multi_host(const int *devices, int ndev, const double *Y, uint32_t N, int d, uint32_t T,
                            const double *m0, const double *C0, const double *F, const double *G,
                            const double *V, const double *W, float df, const char *resampler,
                            const char *distribution, uint32_t B, double scale, uint64_t seed,
                            double *X_out, double *w_out, uint32_t *a_out);

#ifdef __cplusplus
}
#endif
#endif /* CUSMC_HIP_H */
