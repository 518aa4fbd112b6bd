// R exports MVN() and MVNPDF() over libcusmc_hip -- replaces src/mvn_dist.rcpp.cpp of the
// reference (same export names, arity and meaning; registered symbols _CuSMC_MVN (2) and
// _CuSMC_MVNPDF (3), src/RcppExports.cpp:105-113).
#include "glue.hpp"

using namespace cusmc_glue;

// [[Rcpp::depends(RcppEigen)]]

//' MultiVariateNormal Distribution
//'
//' @param  mu     [vector]: Mean vector.
//' @param  sigma  [matrix]: Covariance matrix.
//' @return draws  [vector]: one draw from N(mu, sigma)
//' @export
// [[Rcpp::export]]
Eigen::VectorXd MVN(Eigen::VectorXd mu, Eigen::MatrixXd sigma)
{
  // reference: MVN.sample(draws, sigma, 200) -- sigma ITSELF as the square-root factor and a
  // 3x variance inflation (src/mvn_dist.rcpp.cpp:35).  Default here: the
  // statistically correct draw with Q = eigen square root; options(CuSMC.compat = TRUE)
  // restores the reference's distribution.
  const int d = (int)mu.size();
  const bool compat = Rcpp::as<bool>(Rcpp::Function("getOption")("CuSMC.compat", false));
  RowMatrix Q(d, d);
  if (compat) {
    Q = sigma;
  } else {
    const RowMatrix s = sigma;
    check(cusmc_eigen_sqrt(s.data(), d, Q.data()));
  }
  const Stream st = next_stream();
  Eigen::VectorXd draws(d);
  check(cusmc_sample_host(context(), CUSMC_MVN, 0.f, mu.data(), Q.data(), d, compat ? std::sqrt(3.0) : 1.0,
                          st.seed, st.call, 1, draws.data()));
  return draws;
}

//' MultiVariateNormal Probability Density Function
//'
//' @param x      [vector | d x N matrix]: point, or particles in columns.
//' @param mu     [vector]: Mean vector.
//' @param sigma  [matrix]: Covariance matrix.
//' @return       [numeric]: density (one value per column of x)
//' @export
// [[Rcpp::export]]
Eigen::VectorXd MVNPDF(Eigen::MatrixXd x, Eigen::VectorXd mu, Eigen::MatrixXd sigma)
{
  // reference: F = I; MVN(mu, sigma).pdf(x, F)  (src/mvn_dist.rcpp.cpp:52-58).  A d x N
  // column-major matrix IS the ABI's N x d row-major batch: no repacking.
  if (x.rows() != mu.size()) Rcpp::stop("x has %d rows, mu has %d entries", (int)x.rows(), (int)mu.size());
  Dist dist(CUSMC_MVN, &mu, sigma, 0.f);
  Eigen::VectorXd out(x.cols());
  check(cusmc_dist_pdf_host(dist.h, x.data(), x.cols(), x.rows(), nullptr, CUSMC_OUT_DENSITY, out.data()));
  return out;
}
