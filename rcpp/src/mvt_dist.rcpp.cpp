// R exports MVT() and MVTPDF() over libcusmc_hip -- replaces src/mvt_dist.rcpp.cpp
// (registered symbols _CuSMC_MVT (3), _CuSMC_MVTPDF (4)).
#include "glue.hpp"

using namespace cusmc_glue;

// [[Rcpp::depends(RcppEigen)]]

//' MultiVariate T Distribution
//'
//' @param mu     [vector]: Location vector.
//' @param sigma  [matrix]: Dispersion matrix.
//' @param nu     [float]: degrees of freedom.
//' @return draws [vector]: one draw
//' @export
// [[Rcpp::export]]
Eigen::VectorXd MVT(Eigen::VectorXd mu, Eigen::MatrixXd sigma, float nu)
{
  // reference: Q = eigenvectors * sqrt(eigenvalues), MVT.sample(draws, Q, 200), with one
  // sqrt(nu/chi2) PER COMPONENT (src/mvt_dist.rcpp.cpp:35-47, src/statistics.cc.cpp:385-386,411)
  const int d = (int)mu.size();
  if (sigma.rows() != d || sigma.cols() != d) Rcpp::stop("sigma must be %d x %d", d, d);
  const bool compat = Rcpp::as<bool>(Rcpp::Function("getOption")("CuSMC.compat", false));
  const RowMatrix &Q = session().eigen_sqrt(sigma);
  Eigen::VectorXd draws(d);
  check(cusmc_sample_host(context(), CUSMC_MVT, nu, mu.data(), Q.data(), d, compat ? std::sqrt(3.0) : 1.0,
                          next_key(), 0, 1, draws.data()));
  return draws;
}

//' MultiVariate T Probability Density Function
//'
//' @param x      [vector | d x N matrix]: one point (returns one number, as the reference), or particles in columns.
//' @param mu     [vector]: Location vector.
//' @param sigma  [matrix]: Dispersion matrix.
//' @param nu     [float]: degrees of freedom.
//' @return       [numeric]: density (one value per column of x)
//' @export
// [[Rcpp::export]]
SEXP MVTPDF(SEXP x, Eigen::VectorXd mu, Eigen::MatrixXd sigma, float nu)
{
  // reference: double MVTPDF(VectorXd x, ...)  (src/mvt_dist.rcpp.cpp:60-66); x as in MVNPDF
  return density(session().distribution(CUSMC_MVT, mu, sigma, nu), x, (int)mu.size());
}
