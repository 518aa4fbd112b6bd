// R export run() over libcusmc_hip -- replaces src/run.rcpp.cpp:58-126 and, below it,
// particle_filter() / initialize() / MCMC() / propagate_K() / reweight_G()
// (src/particle_filter.cpp, src/mcmc.cpp) with ONE device-resident call
// (registered symbol _CuSMC_run (14)).
#include <fstream>
#include <string>
#include <vector>

#include "glue.hpp"

using namespace cusmc_glue;
using Rcpp::List;

// [[Rcpp::depends(RcppEigen)]]

// writeOutput -- src/io.cpp:7-43: y_t.csv and x_t_N<p>.csv in the working directory
static void writeOutput(const RowMatrix &Yt, const std::vector<double> &w, const std::vector<double> &X,
                        unsigned N, unsigned d, unsigned T, unsigned p)
{
  Rcpp::Rcout << "\nWriting output!\n";
  std::ofstream fy("y_t.csv"), fx("x_t_N" + std::to_string(p) + ".csv");
  fy << "y" << std::endl;
  fx << "w,x" << std::endl;
  for (unsigned i = 0; i < T; ++i) {
    fx << w[(size_t)i * N];
    for (unsigned j = 0; j < d; ++j) {
      fy << Yt(i, j) << ",";
      fx << "," << X[((size_t)i * N + p) * d + j];
    }
    fy << std::endl;
    fx << std::endl;
  }
}

//' Run simulations
//'
//' @param N            [integer]: Number of particles.
//' @param d            [integer]: Number of parameters.
//' @param timeSteps    [integer]: Total time steps.
//' @param Y            [matrix]: Input observations, d x timeSteps (columns = time).
//' @param m0           [vector]: Initial mean at t=0
//' @param C0           [matrix]: Initial covariance at t=0
//' @param F            [matrix]: Observation matrix
//' @param G            [matrix]: Transition matrix
//' @param V            [matrix]: Observation noise covariance
//' @param W            [matrix]: State noise covariance
//' @param df           [float]:  Degrees of freedom (for MVT)
//' @param resampler    [string]: "metropolis"
//' @param distribution [string]: "mvn" or "mvt"
//' @param p            [integer]: particle written to x_t_N<p>.csv
//' @export
// [[Rcpp::export]]
List run(unsigned &N, unsigned &d, unsigned &timeSteps, Eigen::MatrixXd Y, Eigen::VectorXd m0,
         Eigen::MatrixXd C0, Eigen::MatrixXd F, Eigen::MatrixXd G, Eigen::MatrixXd V, Eigen::MatrixXd W,
         float df, std::string resampler, std::string distribution, unsigned p = 0)
{
  if (p >= N) Rcpp::stop("p = %u must be < N = %u", p, N);  // assert(p < N): src/run.rcpp.cpp:64
  if (Y.rows() != (int)d || Y.cols() != (int)timeSteps) Rcpp::stop("Y must be d x timeSteps");
  const RowMatrix Yt = Y.transpose();  // row t = y_t (the reference stores Y.col(t): :91)
  const RowMatrix C0r = C0, Fr = F, Gr = G, Vr = V, Wr = W;
  const uint64_t key = next_key();  // a fresh Philox key per call: repeated run()s are independent replications
  std::vector<double> X((size_t)timeSteps * N * d), w((size_t)timeSteps * N);

  Rcpp::Rcout << "Simulating... " << std::flush;  // src/particle_filter.cpp:30
  // df goes to the filter AS df: the reference passes runtime/resampler/distribution/df in the
  // wrong slots (src/run.rcpp.cpp:100-106 vs inst/include/particle_filter.hpp:12-19).
  // B = 10 is mcmc.cpp:291.  With CUSMC_DEVICES="0,1,..." in the environment the same call shards the
  // particles over those GPUs below the C ABI (include/cusmc_hip.h: cusmc_pf_run_multi_host).
  check(cusmc_pf_run_host(context(), Yt.data(), N, (int)d, timeSteps, m0.data(), C0r.data(), Fr.data(),
                          Gr.data(), Vr.data(), Wr.data(), df, resampler.c_str(), distribution.c_str(), 10,
                          1.0, key, X.data(), w.data(), nullptr));
  Rcpp::Rcout << "Done." << std::endl;
  writeOutput(Yt, w, X, N, d, timeSteps, p);

  // weights: T x N matrix; posterior_x: T x N x d array (R arrays are column-major)
  Rcpp::NumericMatrix weights(timeSteps, N);
  Rcpp::NumericVector theta((size_t)timeSteps * N * d);
  for (unsigned t = 0; t < timeSteps; ++t)
    for (unsigned i = 0; i < N; ++i) {
      weights(t, i) = w[(size_t)t * N + i];
      for (unsigned k = 0; k < d; ++k)
        theta[t + (size_t)timeSteps * (i + (size_t)N * k)] = X[((size_t)t * N + i) * d + k];
    }
  theta.attr("dim") = Rcpp::IntegerVector::create((int)timeSteps, (int)N, (int)d);
  return List::create(Rcpp::Named("weights") = weights, Rcpp::Named("posterior_x") = theta);
}
