// glue.hpp -- Rcpp/RcppEigen <-> include/cusmc_hip.h.
//
// The only code that knows both worlds.  Everything numerical is below the C ABI; this header
// converts Eigen objects to the ABI's plain row-major buffers and ABI status codes to R errors.
// The reference's equivalent convention is CUDA_CALL -> Rprintf + Rcpp::stop("EXIT_FAILURE")
// (inst/include/support.cuh:9-14); here the message of cusmc_last_error() becomes the R error.
//
// NOT COMPILED IN THE BUILD CONTAINER (no R, Rcpp or Eigen there): see INTEGRATION.md.
#ifndef CUSMC_GLUE_HPP
#define CUSMC_GLUE_HPP

#include <RcppEigen.h>

#include <cstdlib>
#include <cstring>
#include <list>
#include <random>
#include <vector>

#include <cusmc_hip.h>

namespace cusmc_glue {

typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMatrix;

inline void check(int status)
{
  if (status != CUSMC_OK) Rcpp::stop("CuSMC (HIP): %s", cusmc_last_error());
}

// One context per R session (R calls on its single main thread).  Heap objects that are released by
// R_unload_CuSMC (RcppExports.cpp) when the package is unloaded and otherwise left to the process exit:
// no destructor of this library ever runs after the HIP runtime's own static teardown.
struct Session;
Session &session();

// Seed for the draw / resample / run exports: CUSMC_SEED if set, else the OS (the reference draws a
// fresh std::random_device seed per call: src/samplers.cpp:10-11).  Every call gets a Philox key of
// its own, cusmc_stream_key(seed, call counter): successive calls are independent replications and
// never share counters, and CUSMC_SEED reproduces the whole sequence of calls of a session.
inline uint64_t next_key();

// A distribution object of the reference's Distributions registry (src/mcmc.cpp:53-58) with the
// host values it was built from; owned by the session's cache.
struct Dist {
  cusmc_dist *h = nullptr;
  int kind = CUSMC_MVN;
  float nu = 0.f;
  std::vector<double> mu, sigma;  // sigma as passed (column-major; symmetric, so also row-major)
  Dist(int kind_, const Eigen::VectorXd &mu_, const Eigen::MatrixXd &sigma_, float nu_);
  ~Dist() { cusmc_dist_destroy(h); }
  Dist(const Dist &) = delete;
  Dist &operator=(const Dist &) = delete;
  bool matches(int k, const Eigen::VectorXd &m, const Eigen::MatrixXd &s, float n) const
  {
    return k == kind && (k == CUSMC_MVN || n == nu) && (size_t)m.size() == mu.size() &&
           (size_t)s.size() == sigma.size() && !std::memcmp(m.data(), mu.data(), mu.size() * 8) &&
           !std::memcmp(s.data(), sigma.data(), sigma.size() * 8);
  }
};

struct Session {
  static const size_t kKeep = 4;
  cusmc_ctx *ctx = nullptr;
  uint64_t seed = 0, calls = 0;
  // The R-level density calls build a distribution object per call (as the reference does:
  // src/mvn_dist.rcpp.cpp:55).  A loop over particles with the same (mu, sigma) -- the reference's
  // typical R usage -- would pay a factorisation, four device allocations and their release per
  // particle, so the last few objects are kept: a repeated call is an upload of x, one launch and 8
  // bytes back.  Likewise the eigen square roots behind MVN() / MVT() draws.
  std::list<Dist> dists;
  struct Sqrt { std::vector<double> sigma; RowMatrix Q; };
  std::list<Sqrt> sqrts;

  Session()
  {
    check(cusmc_ctx_create(-1, &ctx));
    if (const char *env = std::getenv("CUSMC_SEED")) {
      seed = std::strtoull(env, nullptr, 10);
    } else {
      std::random_device rd;
      seed = ((uint64_t)rd() << 32) | rd();
    }
  }
  ~Session()
  {
    dists.clear();  // distribution handles first, then the context they were created on
    cusmc_ctx_destroy(ctx);
  }

  cusmc_dist *distribution(int kind, const Eigen::VectorXd &mu, const Eigen::MatrixXd &sigma, float nu)
  {
    for (auto it = dists.begin(); it != dists.end(); ++it)
      if (it->matches(kind, mu, sigma, nu)) {
        dists.splice(dists.begin(), dists, it);  // most recently used first
        return dists.front().h;
      }
    dists.emplace_front(kind, mu, sigma, nu);  // throws (Rcpp::stop) on a bad sigma: nothing is cached then
    while (dists.size() > kKeep) dists.pop_back();
    return dists.front().h;
  }

  // Q = V sqrt(Lambda) of eigenSolver() (src/linear_algebra.cpp:10-23), row-major
  const RowMatrix &eigen_sqrt(const Eigen::MatrixXd &sigma)
  {
    const size_t n = (size_t)sigma.size();
    for (auto it = sqrts.begin(); it != sqrts.end(); ++it)
      if (it->sigma.size() == n && !std::memcmp(it->sigma.data(), sigma.data(), n * 8)) {
        sqrts.splice(sqrts.begin(), sqrts, it);
        return sqrts.front().Q;
      }
    const int d = (int)sigma.rows();
    Sqrt e;
    e.sigma.assign(sigma.data(), sigma.data() + n);
    e.Q.resize(d, d);
    const RowMatrix s = sigma;
    check(cusmc_eigen_sqrt(s.data(), d, e.Q.data()));
    sqrts.push_front(std::move(e));
    while (sqrts.size() > kKeep) sqrts.pop_back();
    return sqrts.front().Q;
  }
};

inline cusmc_ctx *context() { return session().ctx; }

inline uint64_t next_key()
{
  Session &s = session();
  return cusmc_stream_key(s.seed, ++s.calls);
}

inline Dist::Dist(int kind_, const Eigen::VectorXd &mu_, const Eigen::MatrixXd &sigma_, float nu_)
    : kind(kind_), nu(nu_), mu(mu_.data(), mu_.data() + mu_.size()), sigma(sigma_.data(), sigma_.data() + sigma_.size())
{
  if (sigma_.rows() != sigma_.cols()) Rcpp::stop("sigma must be square");
  if (mu_.size() != sigma_.rows()) Rcpp::stop("mu and sigma differ in dimension");
  const RowMatrix s = sigma_;  // symmetric, but be explicit about the layout
  check(cusmc_dist_create(context(), kind, mu_.data(), s.data(), (int)sigma_.rows(), nu, &h));
}

// The density exports' `x`: the reference takes a length-d vector and returns one number
// (src/mvn_dist.rcpp.cpp:52, src/mvt_dist.rcpp.cpp:60; R/RcppExports.R:28-35 calls it with c(0, 0)).
// That call must keep working unchanged, so x arrives as a SEXP and is inspected here: a numeric with no
// `dim` attribute is ONE particle (result: numeric(1)); a d x N matrix is N particles in columns
// (result: numeric(N)) -- column-major d x N is exactly the ABI's N x d row-major batch, no repacking.
inline SEXP density(cusmc_dist *dist, SEXP x, int d)
{
  Rcpp::NumericVector xv(x);  // (coerces integer / logical input to double, as Rcpp's Eigen importer does)
  int64_t n = 1;
  bool batched = false;
  if (Rf_isMatrix(x)) {
    const Rcpp::IntegerVector dim = xv.attr("dim");
    if (dim[0] != d) Rcpp::stop("x has %d rows, mu has %d entries", (int)dim[0], d);
    n = dim[1];
    batched = true;
  } else if (xv.size() != d) {
    Rcpp::stop("x has %d entries, mu has %d", (int)xv.size(), d);
  }
  Rcpp::NumericVector out((R_xlen_t)n);
  check(cusmc_dist_pdf_host(dist, xv.begin(), n, d, nullptr, CUSMC_OUT_DENSITY, out.begin()));
  if (!batched) return Rcpp::wrap((double)out[0]);
  return out;
}

}  // namespace cusmc_glue
#endif
