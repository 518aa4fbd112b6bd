// cusmc_host.hpp -- C++ host-side mirror of the reference's operator interface for the hot
// path, over the C ABI (include/cusmc_hip.h).  Header-only, no Eigen, no Rcpp: the reference's
// toolchain (R + Rcpp + RcppEigen) is absent from the build image, so this is the C++ layer a
// compiled caller links against; the Rcpp glue (rcpp/src) is the same thing with Eigen types.
//
// Same names, argument meaning and error behaviour as
//   StatisticalDistribution / MultiVariateNormalDistribution / MultiVariateTStudentDistribution
//       inst/include/statistics.hpp:36-250   (pdf(y), pdf(y,F), getNorm(), sample(draws,Q,n))
//   Sampler::metropolis_hastings              inst/include/samplers.hpp:7-18, src/samplers.cpp:7-36
//   particle_filter                           inst/include/particle_filter.hpp:12-19
// Errors: the reference dies through Rcpp::stop (inst/include/support.cuh:9-14); here every
// non-zero ABI status throws cusmc::Error carrying cusmc_last_error().
#ifndef CUSMC_HOST_HPP
#define CUSMC_HOST_HPP

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cusmc_hip.h"

namespace cusmc {

typedef unsigned dim_t;  // inst/include/types.hpp:12

struct Error : std::runtime_error {
  int code;
  Error(int c, const char *msg) : std::runtime_error(msg), code(c) {}
};
inline void check(int status)
{
  if (status != CUSMC_OK) throw Error(status, cusmc_last_error());
}

typedef std::vector<double> Vector;  // Eigen::VectorXd

// Eigen::MatrixXd stand-in: column-major like Eigen, so (i, j) indexing reads the same.
struct Matrix {
  int rows = 0, cols = 0;
  std::vector<double> data;
  Matrix() {}
  Matrix(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.0) {}
  double &operator()(int i, int j) { return data[(size_t)j * rows + i]; }
  double operator()(int i, int j) const { return data[(size_t)j * rows + i]; }
  static Matrix Identity(int n)
  {
    Matrix m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1.0;
    return m;
  }
  std::vector<double> row_major() const  // what the ABI takes
  {
    std::vector<double> r((size_t)rows * cols);
    for (int i = 0; i < rows; ++i)
      for (int j = 0; j < cols; ++j) r[(size_t)i * cols + j] = (*this)(i, j);
    return r;
  }
};

class Context {
 public:
  explicit Context(int device = -1) { check(cusmc_ctx_create(device, &h_)); }
  ~Context() { cusmc_ctx_destroy(h_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  cusmc_ctx *get() const { return h_; }
  static Context &instance()
  {
    static Context ctx;
    return ctx;
  }

 private:
  cusmc_ctx *h_ = nullptr;
};

// eigenSolver(I_sol, sigma) -- src/linear_algebra.cpp:10-23
inline void eigenSolver(Matrix &I_sol, const Matrix &sigma)
{
  const std::vector<double> s = sigma.row_major();
  std::vector<double> q(s.size());
  check(cusmc_eigen_sqrt(s.data(), sigma.rows, q.data()));
  I_sol = Matrix(sigma.rows, sigma.cols);
  for (int i = 0; i < sigma.rows; ++i)
    for (int j = 0; j < sigma.cols; ++j) I_sol(i, j) = q[(size_t)i * sigma.cols + j];
}

class StatisticalDistribution {
 public:
  virtual ~StatisticalDistribution() { cusmc_dist_destroy(h_); }
  StatisticalDistribution(const StatisticalDistribution &) = delete;
  StatisticalDistribution &operator=(const StatisticalDistribution &) = delete;

  // pdf(y): the quadratic form is in y itself, mu is NOT subtracted (src/statistics.cc.cpp:171-180)
  double pdf(const Vector &y) const
  {
    const Vector zero(y.size(), 0.0);
    double out = 0.0;
    check(cusmc_dist_reweight_host(h_, zero.data(), 1, (int64_t)y.size(), y.data(), nullptr,
                                   CUSMC_OUT_DENSITY, &out));
    return out;
  }
  // pdf(y, F): r = y - F mu  (src/statistics.cc.cpp:183-196, :295-311)
  double pdf(const Vector &y, const Matrix &F) const
  {
    const std::vector<double> f = F.row_major();
    double out = 0.0;
    check(cusmc_dist_pdf_host(h_, y.data(), 1, (int64_t)y.size(), f.data(), CUSMC_OUT_DENSITY, &out));
    return out;
  }
  // batched forms: X is N x d row-major (particles contiguous)
  void pdf(const double *X, int64_t N, const Matrix *F, double *out, bool log_density = false) const
  {
    std::vector<double> f;
    if (F) f = F->row_major();
    check(cusmc_dist_pdf_host(h_, X, N, d_, F ? f.data() : nullptr, log_density ? CUSMC_OUT_LOG : CUSMC_OUT_DENSITY, out));
  }
  double getNorm() const  // src/statistics.cc.cpp:205-211, :332-340
  {
    double ln = 0.0;
    check(cusmc_dist_lognorm(h_, &ln));
    return std::exp(ln);
  }
  Vector mean() const { return mu_; }
  // sample(dist_draws, Q, n_iterations) -- src/statistics.cc.cpp:224-259, :355-412.  n_iterations is
  // kept for signature parity: the reference's 200-term sum is one N(0,3) draw (compat = true).
  void sample(Vector &dist_draws, const Matrix &Q, unsigned n_iterations, uint64_t seed, uint32_t step,
              bool compat = false) const
  {
    (void)n_iterations;
    const std::vector<double> q = Q.row_major();
    dist_draws.assign(d_, 0.0);
    check(cusmc_sample_host(Context::instance().get(), kind_, nu_, mu_.data(), q.data(), d_,
                            compat ? std::sqrt(3.0) : 1.0, seed, step, 1, dist_draws.data()));
  }
  cusmc_dist *handle() const { return h_; }

 protected:
  StatisticalDistribution(int kind, const Vector &m, const Matrix &s, float nu)
      : kind_(kind), d_(s.rows), nu_(nu), mu_(m)
  {
    const std::vector<double> sr = s.row_major();
    check(cusmc_dist_create(Context::instance().get(), kind, m.data(), sr.data(), s.rows, nu, &h_));
  }
  int kind_, d_;
  float nu_;
  Vector mu_;
  cusmc_dist *h_ = nullptr;
};

class MultiVariateNormalDistribution : public StatisticalDistribution {
 public:
  MultiVariateNormalDistribution(const Vector &m, const Matrix &s) : StatisticalDistribution(CUSMC_MVN, m, s, 0.f) {}
};

class MultiVariateTStudentDistribution : public StatisticalDistribution {
 public:
  MultiVariateTStudentDistribution(const Vector &m, const Matrix &s, const float &nu)
      : StatisticalDistribution(CUSMC_MVT, m, s, nu) {}
  float dfree() const { return nu_; }
};

class Sampler {
 public:
  // metropolis_hastings(a_t, w_t, N, t, B): reads w_t[t-1], writes a_t[t*N + i] -- src/samplers.cpp:7-36
  static void metropolis_hastings(unsigned *a_t, const Vector *w_t, size_t N, int t, size_t B = 10,
                                  uint64_t seed = 0)
  {
    check(cusmc_metropolis_host(Context::instance().get(), w_t[t - 1].data(), (uint32_t)N, (uint32_t)B, seed,
                                (uint32_t)t, a_t + (size_t)t * N));
  }
};

// particle_filter(...) -- src/particle_filter.cpp:6-39 with the argument order of its DECLARATION
// (inst/include/particle_filter.hpp:12-19); post_x_t is T x N x d flat, w_t T x N, a_t T x N, y_t T x d.
inline void particle_filter(double *post_x_t, double *w_t, unsigned *a_t, const double *y_t, const Matrix &F,
                            const Matrix &G, const Vector &m0, const Matrix &C0, const Matrix &sigmaV,
                            const Matrix &sigmaW, dim_t N, dim_t d, dim_t timeSteps, float df,
                            const std::string &resampler_opt, const std::string &distribution_opt,
                            uint64_t seed = 0, unsigned B = 10)
{
  check(cusmc_pf_run_host(Context::instance().get(), y_t, N, (int)d, timeSteps, m0.data(), C0.row_major().data(),
                          F.row_major().data(), G.row_major().data(), sigmaV.row_major().data(),
                          sigmaW.row_major().data(), df, resampler_opt.c_str(), distribution_opt.c_str(), B, 1.0,
                          seed, post_x_t, w_t, a_t));
}

// The same filter with the particles sharded over several GPUs of the node (cusmc_pf_run_multi_host: one host
// thread + context per device, peer access; the loop being sharded is src/mcmc.cpp:292-308).  A device may be
// listed more than once.  Same numbers as particle_filter().
inline void particle_filter_multi(const std::vector<int> &devices, double *post_x_t, double *w_t, unsigned *a_t,
                                  const double *y_t, const Matrix &F, const Matrix &G, const Vector &m0,
                                  const Matrix &C0, const Matrix &sigmaV, const Matrix &sigmaW, dim_t N, dim_t d,
                                  dim_t timeSteps, float df, const std::string &resampler_opt,
                                  const std::string &distribution_opt, uint64_t seed = 0, unsigned B = 10)
{
  check(cusmc_pf_run_multi_host(devices.data(), (int)devices.size(), y_t, N, (int)d, timeSteps, m0.data(),
                                C0.row_major().data(), F.row_major().data(), G.row_major().data(),
                                sigmaV.row_major().data(), sigmaW.row_major().data(), df, resampler_opt.c_str(),
                                distribution_opt.c_str(), B, 1.0, seed, post_x_t, w_t, a_t));
}

// Philox key of the `call`-th unseeded call of a session (the reference reseeds from std::random_device per call:
// src/samplers.cpp:10-11); pass it as the `seed` of the functions above.
inline uint64_t stream_key(uint64_t seed, uint64_t call) { return cusmc_stream_key(seed, call); }

}  // namespace cusmc
#endif
