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
  if (sigma.rows() != d || sigma.cols() != d) Rcpp::stop("sigma must be %d x %d", d, d);
  const bool compat = Rcpp::as<bool>(Rcpp::Function("getOption")("CuSMC.compat", false));
  const RowMatrix Q = compat ? RowMatrix(sigma) : session().eigen_sqrt(sigma);
  Eigen::VectorXd draws(d);
  check(cusmc_sample_host(context(), CUSMC_MVN, 0.f, mu.data(), Q.data(), d, compat ? std::sqrt(3.0) : 1.0,
                          next_key(), 0, 1, draws.data()));
  return draws;
}

//' MultiVariateNormal Probability Density Function
//'
//' @param x      [vector | d x N matrix]: one point (returns one number, as the reference), or particles in columns.
//' @param mu     [vector]: Mean vector.
//' @param sigma  [matrix]: Covariance matrix.
//' @return       [numeric]: density (one value per column of x)
//' @export
// [[Rcpp::export]]
SEXP MVNPDF(SEXP x, Eigen::VectorXd mu, Eigen::MatrixXd sigma)
{
  // reference: double MVNPDF(VectorXd x, ...): F = I; MVN(mu, sigma).pdf(x, F)  (src/mvn_dist.rcpp.cpp:52-58).
  // x stays a SEXP so that the documented call MVNPDF(c(0, 0), c(0, 0), diag(2)) -- a numeric WITHOUT a dim
  // attribute, which RcppEigen's MatrixXd importer refuses -- returns its one number as before, and a
  // d x N matrix returns N (glue.hpp: density()).  The distribution object comes from the session's cache.
  return density(session().distribution(CUSMC_MVN, mu, sigma, 0.f), x, (int)mu.size());
}
