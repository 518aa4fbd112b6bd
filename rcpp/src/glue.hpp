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

#include <cusmc_hip.h>

namespace cusmc_glue {

typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMatrix;

inline void check(int status)
{
  if (status != CUSMC_OK) Rcpp::stop("CuSMC (HIP): %s", cusmc_last_error());
}

// One context per R session (R calls on its single main thread).
inline cusmc_ctx *context()
{
  static cusmc_ctx *ctx = nullptr;
  if (!ctx) check(cusmc_ctx_create(-1, &ctx));
  return ctx;
}

// Seed for the draw / resample exports: CUSMC_SEED if set, else the OS (the reference draws a
// fresh std::random_device seed per call: src/samplers.cpp:10-11); successive calls advance a
// call counter that is used as the Philox `step`.
struct Stream { uint64_t seed; uint32_t call; };
Stream next_stream();  // defined in samplers.rcpp.cpp

// RAII handle for a distribution object
struct Dist {
  cusmc_dist *h = nullptr;
  Dist(int kind, const Eigen::VectorXd *mu, const Eigen::MatrixXd &sigma, float nu)
  {
    if (sigma.rows() != sigma.cols()) Rcpp::stop("sigma must be square");
    if (mu && mu->size() != sigma.rows()) Rcpp::stop("mu and sigma differ in dimension");
    const RowMatrix s = sigma;  // symmetric, but be explicit about the layout
    check(cusmc_dist_create(context(), kind, mu ? mu->data() : nullptr, s.data(), (int)sigma.rows(), nu, &h));
  }
  ~Dist() { cusmc_dist_destroy(h); }
  Dist(const Dist &) = delete;
  Dist &operator=(const Dist &) = delete;
};

}  // namespace cusmc_glue
#endif
