// R export metropolis_hastings() over libcusmc_hip -- replaces src/samplers.rcpp.cpp and the
// inner loop of src/samplers.cpp (registered symbol _CuSMC_metropolis_hastings (3)).
#include <vector>

#include "glue.hpp"

using namespace cusmc_glue;

// [[Rcpp::depends(RcppEigen)]]

//' Metropolis Hastings Sampler
//'
//' @param  w  [vector]: Weights
//' @param  N  [integer]: Number of weights
//' @param  B  [integer]: Number of accept/reject iterations per particle
//' @return a  [vector]: Ancestors (0-based, as doubles, like the reference)
//' @export
//' @examples
//' CuSMC::metropolis_hastings(c(0, 0), 2, 10)
// [[Rcpp::export]]
Eigen::VectorXd metropolis_hastings(Eigen::VectorXd w, int N, int B)
{
  // reference: Sampler::metropolis_hastings(ancestors, {w}, N, t = 1, B); a[i] = ancestors[N+i]
  // (src/samplers.rcpp.cpp:38-49)
  if (N < 0 || N > w.size()) Rcpp::stop("N = %d exceeds the %d weights given", N, (int)w.size());
  if (B < 0) Rcpp::stop("B must be non-negative");
  std::vector<uint32_t> a((size_t)N);
  check(cusmc_metropolis_host(context(), w.data(), (uint32_t)N, (uint32_t)B, next_key(), 1, a.data()));
  Eigen::VectorXd out(N);
  for (int i = 0; i < N; ++i) out[i] = (double)a[i];
  return out;
}
