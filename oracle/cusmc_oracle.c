/*
 * cusmc_oracle.c -- CPU restatement of CuSMC's per-particle likelihood / proposal /
 * accept-reject hot path.  Plain C11 + OpenMP, no Eigen, no R.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it.  The shipped path
 * (cusmc_amd/csrc -> libcusmc_hip.so) never links or calls anything in oracle/.
 *
 * Parity status
 *   - pdf / getNorm : pinned by the reference's three published known answers
 *     (CuSMC/CuSMC.tex:95-105, :131-142; man/metropolis_hastings.Rd:22-27) and cross-checked
 *     against scipy.stats (tests/golden/make_golden.py).  The reference has no other fixtures
 *     (SURVEY.md F11), and its C++ cannot be compiled here (every TU includes <RcppEigen.h>;
 *     no R / Rcpp / Eigen in the image), so beyond those three values: PARITY UNPINNED.
 *   - RNG-driven functions (resampler, draws): the reference seeds from std::random_device
 *     on every call (src/samplers.cpp:10-11, src/statistics.cc.cpp:231-232) and cannot be
 *     reproduced by anyone.  This file defines the build's counter-based contract
 *     (Philox4x32-10, pinned by the Random123 known-answer vectors) that the HIP kernels
 *     must match bit-for-bit on index sequences.
 *
 * Conventions: all matrices are dense row-major double, M[i*n+j].  Particle batches are
 * N x d row-major with leading dimension ldx (each particle's d doubles contiguous), the
 * flat layout the reference's own device wrappers build (src/mvn_dist.cu.cpp:722-737).
 * nu is a C float end to end, as in the reference (inst/include/statistics.hpp:30,199).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Dense LU with partial pivoting: what Eigen's MatrixXd::determinant() / ::inverse() do for a
 * dynamic-size matrix (PartialPivLU).  Call sites being restated:
 * src/statistics.cc.cpp:176-177,190,193,301,306,317,320.
 * ---------------------------------------------------------------------------------------- */
static int lu_factor(double *A, int *piv, int n, int *sign)
{
  *sign = 1;
  int singular = 0;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[i * n + k]);
      if (v > best) { best = v; p = i; }
    }
    piv[k] = p;
    if (p != k) {
      for (int j = 0; j < n; ++j) {
        double tmp = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = tmp;
      }
      *sign = -*sign;
    }
    double pivot = A[k * n + k];
    if (pivot == 0.0) { singular = 1; continue; }
    for (int i = k + 1; i < n; ++i) {
      double l = A[i * n + k] / pivot;
      A[i * n + k] = l;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
    }
  }
  return singular;
}

ORACLE_API double oracle_det(const double *S, int n)
{
  double *A = (double *)malloc(sizeof(double) * n * n);
  int *piv = (int *)malloc(sizeof(int) * n);
  int sign;
  memcpy(A, S, sizeof(double) * n * n);
  lu_factor(A, piv, n, &sign);
  double det = (double)sign;
  for (int i = 0; i < n; ++i) det *= A[i * n + i];
  free(A); free(piv);
  return det;
}

ORACLE_API int oracle_inverse(const double *S, double *inv, int n)
{
  double *A = (double *)malloc(sizeof(double) * n * n);
  int *piv = (int *)malloc(sizeof(int) * n);
  double *col = (double *)malloc(sizeof(double) * n);
  int sign;
  memcpy(A, S, sizeof(double) * n * n);
  int singular = lu_factor(A, piv, n, &sign);
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) col[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) { /* apply row swaps */
      int p = piv[k];
      if (p != k) { double t = col[k]; col[k] = col[p]; col[p] = t; }
    }
    for (int i = 0; i < n; ++i) { /* L y = Pb */
      double s = col[i];
      for (int j = 0; j < i; ++j) s -= A[i * n + j] * col[j];
      col[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) { /* U x = y */
      double s = col[i];
      for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * col[j];
      col[i] = s / A[i * n + i];
    }
    for (int i = 0; i < n; ++i) inv[i * n + c] = col[i];
  }
  free(A); free(piv); free(col);
  return singular;
}

/* r = y - F*mu  (F may be NULL: the pdf(y) overload, where the caller folded the mean). */
static void residual(const double *y, const double *mu, const double *F, int n, double *r)
{
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    if (F) { for (int j = 0; j < n; ++j) s += F[i * n + j] * mu[j]; }
    r[i] = F ? (y[i] - s) : y[i];
  }
}

/* (r^T * Sinv) * r, left-associated like the Eigen expression
 * `y.transpose() * sigma.inverse() * y` (src/statistics.cc.cpp:177,193). */
static double quadform(const double *r, const double *Sinv, int n)
{
  double q = 0.0;
  for (int j = 0; j < n; ++j) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += r[i] * Sinv[i * n + j];
    q += s * r[j];
  }
  return q;
}

/* MultiVariateNormalDistribution::getNorm -- src/statistics.cc.cpp:205-211 */
ORACLE_API double oracle_mvn_norm(const double *sigma, int n)
{
  const double sqrt2pi = sqrt(2 * M_PI);
  return 1.0 / (pow(sqrt2pi, (double)n) * pow(oracle_det(sigma, n), 0.5));
}

/* MultiVariateNormalDistribution::pdf(y, F) -- src/statistics.cc.cpp:183-196;
 * with F == NULL the pdf(y) overload, :171-180.  Determinant and inverse are recomputed on
 * every call, exactly as the reference does. */
ORACLE_API double oracle_mvn_pdf(const double *y, const double *mu, const double *sigma,
                                 const double *F, int n)
{
  double *inv = (double *)malloc(sizeof(double) * n * n);
  double *r = (double *)malloc(sizeof(double) * n);
  double norm = oracle_mvn_norm(sigma, n);
  oracle_inverse(sigma, inv, n);
  residual(y, mu, F, n, r);
  double q = quadform(r, inv, n);
  free(inv); free(r);
  return norm * exp(-0.5 * q);
}

/* MultiVariateTStudentDistribution::getNorm -- src/statistics.cc.cpp:332-340.
 * `nu + n` is float arithmetic in the reference (float + unsigned). */
ORACLE_API double oracle_mvt_norm(const double *sigma, int n, float nu)
{
  double pixdf = M_PI * (double)nu;
  double norm1 = pow(pixdf, -0.5 * (double)n) * pow(oracle_det(sigma, n), -0.5);
  float nu_plus_n = nu + (float)n;
  double norm2 = tgamma(0.5 * (double)nu_plus_n) / tgamma(0.5 * (double)nu);
  return norm1 * norm2;
}

/* MultiVariateTStudentDistribution::pdf(y, F) / pdf(y) -- src/statistics.cc.cpp:295-324 */
ORACLE_API double oracle_mvt_pdf(const double *y, const double *mu, const double *sigma,
                                 const double *F, int n, float nu)
{
  double *inv = (double *)malloc(sizeof(double) * n * n);
  double *r = (double *)malloc(sizeof(double) * n);
  double norm = oracle_mvt_norm(sigma, n, nu);
  oracle_inverse(sigma, inv, n);
  residual(y, mu, F, n, r);
  double q1 = quadform(r, inv, n);
  double q = 1.0 + pow((double)nu, -1.0) * q1;
  float nu_plus_n = nu + (float)n;
  free(inv); free(r);
  return norm * pow(q, -0.5 * (double)nu_plus_n);
}

/* Batched, reference-faithful: one fresh distribution + LU det + LU inverse PER PARTICLE,
 * OpenMP parallel-for over particles -- the cost structure of reweight_G's CPU branch
 * (src/mcmc.cpp:193-215) and of R-level MVNPDF/MVTPDF applied particle by particle.
 * dist: 0 = mvn, 1 = mvt.  F may be NULL (identity / pre-folded). */
ORACLE_API void oracle_pdf_batch(const double *X, long N, long ldx, const double *mu,
                                 const double *sigma, const double *F, int n, int dist, float nu,
                                 double *out)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) {
    out[i] = dist ? oracle_mvt_pdf(X + i * ldx, mu, sigma, F, n, nu)
                  : oracle_mvn_pdf(X + i * ldx, mu, sigma, F, n);
  }
}

/* Per-particle covariances: the reference's own pdf() -- which factors the
 * covariance of its distribution object on every call anyway (src/statistics.cc.cpp:176-177, 301,
 * 306) -- applied with a distribution object PER PARTICLE: mu_i (mu + i*ldmu; ldmu == 0: one shared
 * vector; mu == NULL: zero) and sigma_i (sigma + i*n*n).  Densities, like pdf(). */
ORACLE_API void oracle_pdf_percov(const double *X, long N, long ldx, const double *mu, long ldmu,
                                  const double *sigma, int n, int dist, float nu, double *out)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) {
    /* the pdf(y) overload takes the residual itself (it ignores the object's mu: :171-180) */
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    for (int a = 0; a < n; ++a) r[a] = X[i * ldx + a] - (mu ? mu[i * ldmu + a] : 0.0);
    const double *S = sigma + i * (long)n * n;
    out[i] = dist ? oracle_mvt_pdf(r, NULL, S, NULL, n, nu) : oracle_mvn_pdf(r, NULL, S, NULL, n);
    free(r);
  }
}

/* Batched Cholesky, operation for operation what cusmc_amd/csrc/kernels/percov.hip does (row by
 * row, fma chains in index order, one sqrt and one divide per element), so the factors can be
 * compared bit for bit; logdet through libm's log.  info[i] = 0 or 1 + the first bad pivot. */
ORACLE_API void oracle_chol_batched(const double *sigma, long N, int n, double *L, double *logdet, int *info)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) {
    const double *S = sigma + i * (long)n * n;
    double *l = L + i * (long)n * n;
    int bad = 0;
    double ld = 0.0;
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) l[r * n + c] = c <= r ? S[r * n + c] : 0.0;
    for (int j = 0; j < n; ++j) {
      double sum = l[j * n + j];
      for (int k = 0; k < j; ++k) sum = fma(-l[j * n + k], l[j * n + k], sum);
      if (!(sum > 0.0) && bad == 0) bad = j + 1;
      const double ljj = sqrt(sum);
      l[j * n + j] = ljj;
      ld += log(ljj);
      for (int r = j + 1; r < n; ++r) {
        double t = l[r * n + j];
        for (int k = 0; k < j; ++k) t = fma(-l[r * n + k], l[j * n + k], t);
        l[r * n + j] = t / ljj;
      }
    }
    if (logdet) logdet[i] = bad ? NAN : 2.0 * ld;
    if (info) info[i] = bad;
  }
}

/* reweight_G, CPU branch -- src/mcmc.cpp:185-215:
 *   w[i] = pdf_{0,V}( y - F * x_i )     (the pdf(y) overload, mean zero) */
ORACLE_API void oracle_reweight(const double *X, long N, long ldx, const double *y,
                                const double *F, const double *V, int n, int dist, float nu,
                                double *w)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) {
    double *r = (double *)malloc(sizeof(double) * n);
    const double *x = X + i * ldx;
    for (int a = 0; a < n; ++a) {
      double s = 0.0;
      for (int b = 0; b < n; ++b) s += F[a * n + b] * x[b];
      r[a] = y[a] - s;
    }
    w[i] = dist ? oracle_mvt_pdf(r, NULL, V, NULL, n, nu) : oracle_mvn_pdf(r, NULL, V, NULL, n);
    free(r);
  }
}

/* ------------------------------------------------------------------------------------------
 * Hoisted-factor form (factor Sigma once, triangular solve per particle).  Not how the
 * reference spends its time, but the same function of the inputs; used (i) as the second CPU
 * baseline line that separates algorithmic gain from hardware gain and
 * (ii) as the checker at sizes where the O(N d^3) faithful form is too slow.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int oracle_cholesky(const double *S, double *L, int n)
{
  memset(L, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; ++j) {
    double s = S[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0.0)) return j + 1;
    double ljj = sqrt(s);
    L[j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double t = S[i * n + j];
      for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / ljj;
    }
  }
  return 0;
}

/* dist 0: log N(x; mu, Sigma);  dist 1: log t_nu(x; mu, Sigma).  F NULL => r = x - mu,
 * else r = x - F mu (pdf(y,F) semantics).  Returns nonzero if Sigma is not SPD. */
ORACLE_API int oracle_logpdf_hoisted(const double *X, long N, long ldx, const double *mu,
                                     const double *sigma, const double *F, int n, int dist,
                                     float nu, double *out)
{
  double *L = (double *)malloc(sizeof(double) * n * n);
  double *m = (double *)malloc(sizeof(double) * n);
  int rc = oracle_cholesky(sigma, L, n);
  if (rc) { free(L); free(m); return rc; }
  double logdet = 0.0;
  for (int i = 0; i < n; ++i) logdet += 2.0 * log(L[i * n + i]);
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    if (F) { for (int j = 0; j < n; ++j) s += F[i * n + j] * mu[j]; } else s = mu ? mu[i] : 0.0;
    m[i] = s;
  }
  float nu_plus_n = nu + (float)n;
  double lognorm;
  if (dist == 0)
    lognorm = -0.5 * ((double)n * log(2 * M_PI) + logdet);
  else
    lognorm = lgamma(0.5 * (double)nu_plus_n) - lgamma(0.5 * (double)nu) -
              0.5 * (double)n * log(M_PI * (double)nu) - 0.5 * logdet;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) {
    double z[n];
    const double *x = X + i * ldx;
    double q = 0.0;
    for (int a = 0; a < n; ++a) {
      double s = x[a] - m[a];
      for (int b = 0; b < a; ++b) s -= L[a * n + b] * z[b];
      z[a] = s / L[a * n + a];
      q += z[a] * z[a];
    }
    out[i] = dist ? lognorm - 0.5 * (double)nu_plus_n * log1p(q / (double)nu) : lognorm - 0.5 * q;
  }
  free(L); free(m);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG contract of the build: Philox4x32-10 (Salmon et al.,
 * SC'11; Random123).  Pinned by the Random123 known-answer vectors in tests/test_oracle.py.
 *   key     = (seed_lo, seed_hi)
 *   counter = (index, sub, step, domain)
 * domain tags: 1 resampler, 2 proposal normals, 3 chi-square normals, 4 initial normals,
 *              5 chi-square accept/boost uniforms, 6 chi-square closed-form uniforms, 8 the closed form's normals
 *              for odd nu (contract 4: see
 *              chi_square_for), 7 resampler accept refinement, 16 + q resampler index redraws (contract 3).
 * ---------------------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

ORACLE_API void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in [0,1) from two words, and an unbiased-to-2^-64 integer in [0,N). */
static inline double u01_53(uint32_t hi, uint32_t lo)
{
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * 0x1.0p-53;
}
static inline uint32_t uint_below(uint32_t hi, uint32_t lo, uint32_t N)
{
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(((unsigned __int128)v * (unsigned __int128)N) >> 64);
}

/* Sampler::metropolis_hastings -- src/samplers.cpp:7-36, inner loop :21-35.
 * Per particle i: k = i; repeat B times: u ~ U[0,1) THEN j ~ UnifInt{0..N-1} (that draw
 * order); accept iff (u <= w[j] / w[k]) -- the same division and <=, so a NaN ratio never
 * accepts (man/metropolis_hastings.Rd:22-27).  u, j, k are thread-private here (the
 * reference's shared declarations are a data race).
 * `step` plays the role of the reference's t (a_t[t*N+i] = k); the caller owns the offset.
 *
 * RNG CONTRACT 3 (round 3; contracts 1-2 spent one Philox block per chain step: 53 bits for u, 64 for j).  ONE block
 * serves TWO steps, and both draws stay EXACT by taking more bits only when the first 32 cannot decide:
 *   block (i, n / 2, step, 1), half h = n % 2:  a = word 2h,  b = word 2h + 1.
 *   u  is a uniform real on [0, 1) whose leading 32 bits are a:  u in [a 2^-32, (a + 1) 2^-32).  With
 *      r = w[j] / w[k] (the reference's division):  (a + 1) 2^-32 <= r  accepts,  a 2^-32 > r  rejects, and only when r
 *      lies inside u's cell (probability 2^-32 per step) are the next 53 bits read: V = u01_53 of words 0, 1 of block
 *      (i, n, step, 7), accept iff V <= r 2^32 - a (exact: the cell's offset).  A NaN ratio rejects.  This IS the
 *      reference's test u <= w[j] / w[k], with an 85-bit u instead of a 53-bit one.
 *   j  is uniform on {0 .. N-1} WITHOUT bias (Lemire's multiply-and-reject): m = b N, j = m >> 32, unless the low word
 *      of m is below tN = (2^32 - N) mod N (probability < N 2^-32): then b is redrawn -- words 0, 1, 2, 3 of block
 *      (i, n, step, 16), then of (i, n, step, 17), ... -- until it is not.  The reference draws
 *      std::uniform_int_distribution (exact as well).
 * One Philox block per two steps is where the resampler's time goes (~50 of its ~75 VALU instructions per step were
 * the ten rounds): profiles/r03_pmc_mh.md. */
static inline uint32_t mh_tn(uint32_t N) { return (uint32_t)(0u - N) % N; }

static inline uint32_t mh_index(uint32_t b, uint32_t i, uint32_t n, uint32_t step, const uint32_t key[2], uint32_t N,
                                uint32_t tN)
{
  uint64_t m = (uint64_t)b * N;
  if ((uint32_t)m < tN) {
    for (uint32_t q = 0;; ++q) {
      uint32_t ctr[4] = {i, n, step, 16u + q}, r[4];
      oracle_philox4x32_10(ctr, key, r);
      int found = 0;
      for (int c = 0; c < 4 && !found; ++c) {
        m = (uint64_t)r[c] * N;
        found = (uint32_t)m >= tN;
      }
      if (found) break;
    }
  }
  return (uint32_t)(m >> 32);
}

static inline int mh_accept(uint32_t a, double r, uint32_t i, uint32_t n, uint32_t step, const uint32_t key[2])
{
  if (r != r) return 0;
  const double lo = (double)a * 0x1.0p-32, hi = lo + 0x1.0p-32; /* both exact */
  if (hi <= r) return 1;
  if (lo > r) return 0;
  uint32_t ctr[4] = {i, n, step, 7u}, x[4];
  oracle_philox4x32_10(ctr, key, x);
  return u01_53(x[0], x[1]) <= r * 0x1.0p32 - (double)a; /* r 2^32 in [a, a + 1]: the difference is exact */
}

/* chains [first, first + count) of the N: a[i - first] (what one rank of a sharded resample computes) */
ORACLE_API void oracle_metropolis_range(uint32_t *a, const double *w, uint32_t N, uint32_t B,
                                        uint64_t seed, uint32_t step, uint32_t first, uint32_t count)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t tN = mh_tn(N);
#pragma omp parallel for schedule(static)
  for (uint32_t c = 0; c < count; ++c) {
    const uint32_t i = first + c;
    uint32_t k = i;
    double wk = w[k];
    uint32_t r[4] = {0, 0, 0, 0};
    for (uint32_t n = 0; n < B; ++n) {
      if ((n & 1u) == 0) {
        uint32_t ctr[4] = {i, n >> 1, step, 1u};
        oracle_philox4x32_10(ctr, key, r);
      }
      const uint32_t ua = r[2 * (n & 1u)];
      const uint32_t j = mh_index(r[2 * (n & 1u) + 1], i, n, step, key, N, tN);
      double wj = w[j];
      if (mh_accept(ua, wj / wk, i, n, step, key)) { k = j; wk = wj; }
    }
    a[c] = k;
  }
}

ORACLE_API void oracle_metropolis(uint32_t *a, const double *w, uint32_t N, uint32_t B,
                                  uint64_t seed, uint32_t step)
{
  oracle_metropolis_range(a, w, N, B, seed, step, 0u, N);
}

/* exp(t) for t <= 0 as ONE fixed sequence of correctly rounded operations (reduction t = k ln2 + r,
 * fdlibm e_exp.c's rational form; this file is built with -ffp-contract=off), mirrored operation for
 * operation by cusmc_amd/csrc/kernels/smallops.h:exp_nonpos -- so that the log-weight accept test is
 * bit-identical on both sides, which libm's exp and the GPU library's would not be. */
static double exp_nonpos(double t)
{
  if (!(t > -746.0)) return t != t ? t : 0.0;
  const double kf = rint(t * 1.44269504088896338700e+00);
  double r = fma(-kf, 6.93147180369123816490e-01, t);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  const double r2 = r * r;
  const double pp = fma(r2, fma(r2, fma(r2, fma(r2, 4.13813679705723846039e-08, -1.65339022054652515390e-06),
                                        6.61375632143793436117e-05), -2.77777777770155933842e-03),
                        1.66666666666666019037e-01);
  const double c = fma(-r2, pp, r);
  const double num = r * c;
  const double den = c - 2.0;
  const double quo = num / den;
  const double e = 1.0 - (quo - r);
  return ldexp(e, (int)kf);
}

/* The Metropolis chain over log-weights: the reference's test (src/samplers.cpp:27-31) with
 * w = exp(lw), i.e. u <= exp(lw[j] - lw[k]); a non-negative difference always accepts (u < 1).  Draws as above. */
ORACLE_API void oracle_metropolis_log(uint32_t *a, const double *lw, uint32_t N, uint32_t B,
                                      uint64_t seed, uint32_t step)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t tN = mh_tn(N);
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < N; ++i) {
    uint32_t k = i;
    double lk = lw[k];
    uint32_t r[4] = {0, 0, 0, 0};
    for (uint32_t n = 0; n < B; ++n) {
      if ((n & 1u) == 0) {
        uint32_t ctr[4] = {i, n >> 1, step, 1u};
        oracle_philox4x32_10(ctr, key, r);
      }
      const uint32_t ua = r[2 * (n & 1u)];
      const uint32_t j = mh_index(r[2 * (n & 1u) + 1], i, n, step, key, N, tN);
      double lj = lw[j], t = lj - lk;
      if (t >= 0.0 || mh_accept(ua, exp_nonpos(t), i, n, step, key)) { k = j; lk = lj; }
    }
    a[i] = k;
  }
}

ORACLE_API double oracle_exp_nonpos(double t) { return exp_nonpos(t); }

/* Standard-normal pair by Box-Muller from one Philox block: u1 in (0,1], u2 in [0,1). */
static inline void normal_pair(const uint32_t r[4], double *z0, double *z1)
{
  double u1 = 1.0 - u01_53(r[0], r[1]);
  double u2 = u01_53(r[2], r[3]);
  double rad = sqrt(-2.0 * log(u1));
  double ang = 2.0 * M_PI * u2;
  *z0 = rad * cos(ang);
  *z1 = rad * sin(ang);
}

/* xi_j, j < d, for (particle, step): block (index=particle, sub=j/2, step, domain). */
static void normals_for(uint32_t particle, uint32_t step, uint32_t domain, const uint32_t key[2],
                        int d, double *xi)
{
  for (int j = 0; j < d; j += 2) {
    uint32_t ctr[4] = {particle, (uint32_t)(j >> 1), step, domain}, r[4];
    double z0, z1;
    oracle_philox4x32_10(ctr, key, r);
    normal_pair(r, &z0, &z1);
    xi[j] = z0;
    if (j + 1 < d) xi[j + 1] = z1;
  }
}

/* chi^2_nu draws, RNG CONTRACT 4 (contract 2 of round 3, closed form extended from nu = 2, 4 to every integer
 * nu <= 16; contract 1 keyed two Philox blocks and a whole Box-Muller pair per
 * attempt and component and threw the second normal and half of the uniforms away -- 3 to 4.6 x the cost of
 * the Normal draw on the GPU).  The reference's draws cannot be reproduced by anyone (std::random_device per
 * call, src/statistics.cc.cpp:360-361; curand_init(dev_seed, ...) with a wall-clock seed on the device path), so
 * what has to hold is the LAW: chi[j] ~ chi^2_nu, independent over components j and of the normals
 * (src/statistics.cc.cpp:366, 383-386: `chi[j] = X2(generator)` per component).  Everything is keyed by the
 * component PAIR p = j / 2, half e = j % 2 -- the unit the proposal normals are keyed by as well:
 *
 *   integer nu <= 16   closed form, no rejection, m = floor(nu / 2):
 *           chi^2_nu = -2 ln(u_1 ... u_m)  [ + z^2 if nu is odd ]
 *       Block b < ceil(m / 2) of the pair: (particle, p + (b << 16), step, 6); half e takes its words (2e, 2e+1):
 *         m = 1:  u = ((w_2e : w_2e+1) >> 12 + 1/2) 2^-52                                   in (0, 1), 52 bits
 *         m >= 2: u_{2b+1} = (w_2e + 1/2) 2^-32,  u_{2b+2} = (w_2e+1 + 1/2) 2^-32 (if <= m)  in (0, 1), 32 bits each,
 *                 multiplied in that order starting from 1.0
 *       odd nu: (z_0, z_1) = Box-Muller of block (particle, p, step, 8); half e adds z_e^2 (one fma).
 *       (32-bit uniforms: at m = 2 the product has 2^64 equally likely values; the largest draw is 91.5 where
 *       the exact law has 6e-19 beyond it, the smallest 4.7e-10 with 3e-20 below it.  nu = 2 and 4 are what
 *       contracts 2 and 3 drew.)
 *   any other nu   Marsaglia-Tsang as the reference's device helper (src/mvt_dist.cu.cpp:20-61; libstdc++'s
 *       gamma_distribution behind chi_squared_distribution on the CPU path is the same algorithm), squeeze
 *       included, a < 1 boost included.  Attempt m < 63 of pair p: block (particle, 64 p + m, step, 3) ->
 *       Box-Muller -> z0 is half 0's normal, z1 half 1's; block (particle, 64 p + m, step, 5) -> words (2e, 2e+1)
 *       -> half e's accept uniform u = 1 - u01 in (0, 1].  The two halves walk their attempts independently
 *       (each stops at its own first accept; what the other half leaves unused is discarded).  a < 1 boost
 *       uniform: block (particle, 64 p + 63, step, 5), words (2e, 2e+1). */
static double chi_square_for(uint32_t particle, uint32_t j, uint32_t step, const uint32_t key[2],
                             float nu)
{
  const uint32_t p = j >> 1, e = j & 1u;
  if (nu >= 1.0f && nu <= 16.0f && nu == (float)(int)nu) {
    const int m = (int)nu / 2, odd = (int)nu & 1;
    double P = 1.0, chi = 0.0;
    if (m == 1) {
      uint32_t ctr[4] = {particle, p, step, 6u}, r[4];
      oracle_philox4x32_10(ctr, key, r);
      const uint64_t v = (((uint64_t)r[2 * e] << 32) | r[2 * e + 1]) >> 12;
      P = ((double)v + 0.5) * 0x1.0p-52;
    } else {
      for (int b = 0; 2 * b < m; ++b) {
        uint32_t ctr[4] = {particle, p + ((uint32_t)b << 16), step, 6u}, r[4];
        oracle_philox4x32_10(ctr, key, r);
        P *= fma((double)r[2 * e], 0x1.0p-32, 0x1.0p-33);
        if (2 * b + 1 < m) P *= fma((double)r[2 * e + 1], 0x1.0p-32, 0x1.0p-33);
      }
    }
    if (m) chi = -2.0 * log(P);
    if (odd) {
      uint32_t ctr[4] = {particle, p, step, 8u}, r[4];
      double zz[2];
      oracle_philox4x32_10(ctr, key, r);
      normal_pair(r, &zz[0], &zz[1]);
      chi = fma(zz[e], zz[e], chi);
    }
    return chi;
  }
  double a = 0.5 * (double)nu;
  double boost = 1.0;
  if (a < 1.0) {
    uint32_t ctr[4] = {particle, p * 64u + 63u, step, 5u}, r[4];
    oracle_philox4x32_10(ctr, key, r);
    boost = pow(1.0 - u01_53(r[2 * e], r[2 * e + 1]), 1.0 / a);
    a += 1.0;
  }
  double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
  double g = dd; /* value if all 63 attempts reject (probability < 1e-60) */
  for (uint32_t m = 0; m < 63u; ++m) {
    uint32_t ctr[4] = {particle, p * 64u + m, step, 3u}, r[4];
    double zz[2];
    oracle_philox4x32_10(ctr, key, r);
    normal_pair(r, &zz[0], &zz[1]);
    const double z = zz[e];
    double v = 1.0 + c * z;
    if (v <= 0.0) continue;
    v = v * v * v;
    ctr[3] = 5u;
    oracle_philox4x32_10(ctr, key, r);
    double u = 1.0 - u01_53(r[2 * e], r[2 * e + 1]);
    /* the reference's squeeze (src/mvt_dist.cu.cpp:45), u < 1 - 0.0331 z^4 as one fixed sequence of roundings
     * (the kernels evaluate the same fma: smallops.h chi_squeeze), then its log test (:48) */
    const double z2 = z * z;
    if (u < fma(-(0.0331 * z2), z2, 1.0) ||
        log(u) < 0.5 * z * z + dd - dd * v + dd * log(v)) { g = dd * v; break; }
  }
  return 2.0 * g * boost;
}

ORACLE_API int oracle_rng_contract(void) { return 4; }

/* chi^2 draws of components [0, d) of `count` particles starting at `first` (tests/test_distributions.py holds
 * them against the exact chi^2_nu law; the kernels never see this) */
ORACLE_API void oracle_chi_square(double *out, uint32_t first, uint32_t count, int d, float nu, uint64_t seed,
                                  uint32_t step)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < count; ++i)
    for (int j = 0; j < d; ++j) out[(size_t)i * d + j] = chi_square_for(first + i, (uint32_t)j, step, key, nu);
}

/* One proposal draw given the location m (d) and the square-root factor Q (d x d, dense).
 *   mvn:  x = Q xi + m                                  src/statistics.cc.cpp:258
 *   mvt:  x = diag(c) (Q xi) + m,  c_j = sqrt(nu/chi2)  src/statistics.cc.cpp:385-386,411
 * xi ~ N(0, s^2 I).  s = 1 is the statistically correct draw (what the reference's device
 * path does, src/mvn_dist.cu.cpp:24-31); s = sqrt(3) ("clt_compat") is the distribution the
 * reference's CPU path actually produces: (sum_{i<200} (z_i+1)/2 - 100)/sqrt(200/12) is
 * exactly N(0,3) (src/statistics.cc.cpp:245-256). */
static void draw_one(uint32_t particle, uint32_t step, uint32_t domain, const uint32_t key[2],
                     const double *m, const double *Q, int d, int dist, float nu, double scale,
                     double *x)
{
  double xi[d];
  normals_for(particle, step, domain, key, d, xi);
  for (int a = 0; a < d; ++a) {
    double s = 0.0;
    for (int b = 0; b < d; ++b) s += Q[a * d + b] * (scale * xi[b]);
    if (dist) {
      double chi = chi_square_for(particle, (uint32_t)a, step, key, nu);
      s *= sqrt((double)nu / chi);
    }
    x[a] = s + m[a];
  }
}

/* initialize() -- src/mcmc.cpp:44-88:  x_0[i] ~ dist(m0).sample(Q0);  w_0 = 1/N.
 * The filter uses step = 0; the R-level MVN()/MVT() draws (src/mvn_dist.rcpp.cpp:31-37,
 * src/mvt_dist.rcpp.cpp:28-49) are the same transform with a per-call step. */
ORACLE_API void oracle_initialize(double *X0, double *w0, uint32_t N, int d, const double *m0,
                                  const double *Q0, int dist, float nu, double scale,
                                  uint64_t seed, uint32_t step)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < N; ++i) {
    draw_one(i, step, 4u, key, m0, Q0, d, dist, nu, scale, X0 + (size_t)i * d);
    w0[i] = 1.0 / (double)N;
  }
}

/* propagate_K(), CPU branch -- src/mcmc.cpp:112-140:
 *   mu_i = G * x_{t-1}[a_i];   x_t[i] ~ dist(mu_i).sample(Q_w) */
ORACLE_API void oracle_propagate(double *Xt, const double *Xprev, const uint32_t *a, uint32_t N,
                                 int d, const double *G, const double *Qw, int dist, float nu,
                                 double scale, uint64_t seed, uint32_t step)
{
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < N; ++i) {
    double m[d];
    const double *xp = Xprev + (size_t)a[i] * d;
    for (int r = 0; r < d; ++r) {
      double s = 0.0;
      for (int c = 0; c < d; ++c) s += G[r * d + c] * xp[c];
      m[r] = s;
    }
    draw_one(i, step, 2u, key, m, Qw, d, dist, nu, scale, Xt + (size_t)i * d);
  }
}

/* Symmetric eigen square root Q = V sqrt(Lambda) -- src/linear_algebra.cpp:10-23, which calls
 * Eigen's SelfAdjointEigenSolver: Householder tridiagonalisation + implicit QL (EISPACK tred2 /
 * tql2 here), eigenvalues ascending as Eigen returns them.  Eigenvector signs are not defined by the
 * reference (nor by Eigen); the convention here -- largest component positive -- makes Q
 * reproducible.  Q Q^T = Sigma. */
ORACLE_API void oracle_eigen_sqrt(const double *S, double *Q, int n)
{
  double *V = (double *)malloc(sizeof(double) * n * n), *d = (double *)malloc(sizeof(double) * n), *e = (double *)malloc(sizeof(double) * n);
  memcpy(V, S, sizeof(double) * n * n);
#define v(i, j) V[(size_t)(j) * n + (i)] /* column-major: the O(n^3) loops walk down columns */
  // ---- tred2: V <- orthogonal transformation to tridiagonal form; d = diagonal, e = sub-diagonal
  for (int j = 0; j < n; ++j) d[j] = v(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
        v(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1], g = sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        v(j, i) = f;
        g = e[j] + v(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) {
          g += v(k, j) * d[k];
          e[k] += v(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) v(k, j) -= (f * e[k] + g * d[k]);
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    v(n - 1, i) = v(i, i);
    v(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = v(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += v(k, i + 1) * v(k, j);
        for (int k = 0; k <= i; ++k) v(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) v(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) {
    d[j] = v(n - 1, j);
    v(n - 1, j) = 0.0;
  }
  v(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- tql2: implicit QL on the tridiagonal matrix, rotations accumulated into V
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    { const double cand = fabs(d[l]) + fabs(e[l]); if (cand > tst1) tst1 = cand; }
    int m = l;
    while (m < n) {
      if (fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m > l) {
      for (int iter = 0; iter < 200; ++iter) {
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = v(k, i + 1);
            v(k, i + 1) = s * v(k, i) + c * h;
            v(k, i) = c * v(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
        if (fabs(e[l]) <= eps * tst1) break;
      }
    }
    d[l] += f;
    e[l] = 0.0;
  }
  // ascending eigenvalues (selection sort, stable for ties), then the sign convention
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j)
      if (d[j] < p) {
        k = j;
        p = d[j];
      }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int r = 0; r < n; ++r) { const double tmp = v(r, i); v(r, i) = v(r, k); v(r, k) = tmp; }
    }
  }
  for (int j = 0; j < n; ++j) {
    int im = 0;
    for (int i = 1; i < n; ++i)
      if (fabs(v(i, j)) > fabs(v(im, j))) im = i;
    const double sg = v(im, j) < 0.0 ? -1.0 : 1.0;
    const double root = d[j] > 0.0 ? sqrt(d[j]) : 0.0;
    for (int i = 0; i < n; ++i) Q[(size_t)i * n + j] = sg * v(i, j) * root;
  }
#undef v
  free(V); free(d); free(e);
}

/* MCMC() time loop -- src/mcmc.cpp:239-309, loop :292-308, with initialize() in front as
 * particle_filter() does (src/particle_filter.cpp:22-36).  B = 10 is the reference's
 * hard-coded value (mcmc.cpp:291) but is a parameter here.
 *   X : T x N x d,  W : T x N (unnormalised pdf values, as the reference stores them),
 *   A : T x N ancestors (row 0 never written, as in the reference), Y : T x d (row t = y_t).
 * hoisted != 0 evaluates weights through the Cholesky form instead of per-particle LU. */
ORACLE_API int oracle_pf_run(double *X, double *W, uint32_t *A, const double *Y, uint32_t N,
                             int d, uint32_t T, const double *m0, const double *C0,
                             const double *F, const double *G, const double *V, const double *Wc,
                             int dist, float nu, uint32_t B, double scale, uint64_t seed,
                             int hoisted)
{
  double *Q0 = (double *)malloc(sizeof(double) * d * d);
  double *Qw = (double *)malloc(sizeof(double) * d * d);
  oracle_eigen_sqrt(C0, Q0, d);
  oracle_eigen_sqrt(Wc, Qw, d);
  oracle_initialize(X, W, N, d, m0, Q0, dist, nu, scale, seed, 0u);
  int rc = 0;
  for (uint32_t t = 1; t < T && !rc; ++t) {
    double *Xt = X + (size_t)t * N * d;
    const double *Xp = X + (size_t)(t - 1) * N * d;
    oracle_metropolis(A + (size_t)t * N, W + (size_t)(t - 1) * N, N, B, seed, t);
    oracle_propagate(Xt, Xp, A + (size_t)t * N, N, d, G, Qw, dist, nu, scale, seed, t);
    if (!hoisted) {
      oracle_reweight(Xt, N, d, Y + (size_t)t * d, F, V, d, dist, nu, W + (size_t)t * N);
    } else {
      /* r_i = y - F x_i  then log-density through the hoisted factor, exponentiated */
      double *R = (double *)malloc(sizeof(double) * (size_t)N * d);
      for (uint32_t i = 0; i < N; ++i)
        for (int a = 0; a < d; ++a) {
          double s = 0.0;
          for (int b = 0; b < d; ++b) s += F[a * d + b] * Xt[(size_t)i * d + b];
          R[(size_t)i * d + a] = Y[(size_t)t * d + a] - s;
        }
      rc = oracle_logpdf_hoisted(R, N, d, NULL, V, NULL, d, dist, nu, W + (size_t)t * N);
      for (uint32_t i = 0; i < N; ++i) W[(size_t)t * N + i] = exp(W[(size_t)t * N + i]);
      free(R);
    }
  }
  free(Q0); free(Qw);
  return rc;
}

ORACLE_API int oracle_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
