// RcppExports.cpp -- the .Call shims and the registration table of CuSMC.so, written by hand.
//
// Rcpp::compileAttributes() would generate this file from the [[Rcpp::export]] tags in *.rcpp.cpp; it
// is kept in the tree so that the package builds (R CMD INSTALL) without that step and so that the
// upper drop-in boundary is an artefact, not an instruction: R's .Call looks up exactly the six symbols
// the reference registers (src/RcppExports.cpp:105-118 of the reference: _CuSMC_MVN (2), _CuSMC_MVNPDF (3),
// _CuSMC_MVT (3), _CuSMC_MVTPDF (4), _CuSMC_run (14), _CuSMC_metropolis_hastings (3)), with dynamic
// lookup switched off.  Differences from the generated form: the density exports keep `x` as a SEXP
// (glue.hpp: density()), and R_unload_CuSMC releases the session's device objects.
//
// NOT COMPILED IN THE BUILD CONTAINER (no R, Rcpp or Eigen there): see INTEGRATION.md.
#include "glue.hpp"

#include <R_ext/Rdynload.h>

#include <string>

// ---- the export bodies (mvn_dist / mvt_dist / samplers / run .rcpp.cpp) -----------------------------
Eigen::VectorXd MVN(Eigen::VectorXd mu, Eigen::MatrixXd sigma);
SEXP MVNPDF(SEXP x, Eigen::VectorXd mu, Eigen::MatrixXd sigma);
Eigen::VectorXd MVT(Eigen::VectorXd mu, Eigen::MatrixXd sigma, float nu);
SEXP MVTPDF(SEXP x, Eigen::VectorXd mu, Eigen::MatrixXd sigma, float nu);
Rcpp::List run(unsigned &N, unsigned &d, unsigned &timeSteps, Eigen::MatrixXd Y, Eigen::VectorXd m0,
               Eigen::MatrixXd C0, Eigen::MatrixXd F, Eigen::MatrixXd G, Eigen::MatrixXd V, Eigen::MatrixXd W,
               float df, std::string resampler, std::string distribution, unsigned p);
Eigen::VectorXd metropolis_hastings(Eigen::VectorXd w, int N, int B);

namespace cusmc_glue {
static Session *g_session = nullptr;
Session &session()
{
  if (!g_session) g_session = new Session();
  return *g_session;
}
}  // namespace cusmc_glue

namespace {
// SEXP -> owned C++ value, as the generated shims do it (arguments are COPIED: SURVEY.md 8b "Ownership")
template <typename T>
using arg = typename Rcpp::traits::input_parameter<T>::type;
typedef Eigen::VectorXd Vec;
typedef Eigen::MatrixXd Mat;
}  // namespace

// Every shim: exceptions -> R errors (BEGIN_RCPP / END_RCPP), an RNGScope like the reference's (R's RNG
// is not used on either side: set.seed() has no effect, CUSMC_SEED does), the result wrapped into a
// fresh R object.
#define CUSMC_SHIM(expr)    \
  BEGIN_RCPP                \
  Rcpp::RObject result;     \
  Rcpp::RNGScope rng_scope; \
  result = Rcpp::wrap(expr); \
  return result;            \
  END_RCPP

RcppExport SEXP _CuSMC_MVN(SEXP mu, SEXP sigma)
{
  CUSMC_SHIM(MVN(arg<Vec>(mu), arg<Mat>(sigma)))
}

RcppExport SEXP _CuSMC_MVNPDF(SEXP x, SEXP mu, SEXP sigma)
{
  CUSMC_SHIM(MVNPDF(x, arg<Vec>(mu), arg<Mat>(sigma)))
}

RcppExport SEXP _CuSMC_MVT(SEXP mu, SEXP sigma, SEXP nu)
{
  CUSMC_SHIM(MVT(arg<Vec>(mu), arg<Mat>(sigma), arg<float>(nu)))
}

RcppExport SEXP _CuSMC_MVTPDF(SEXP x, SEXP mu, SEXP sigma, SEXP nu)
{
  CUSMC_SHIM(MVTPDF(x, arg<Vec>(mu), arg<Mat>(sigma), arg<float>(nu)))
}

RcppExport SEXP _CuSMC_run(SEXP N, SEXP d, SEXP timeSteps, SEXP Y, SEXP m0, SEXP C0, SEXP F, SEXP G, SEXP V,
                           SEXP W, SEXP df, SEXP resampler, SEXP distribution, SEXP p)
{
  BEGIN_RCPP
  Rcpp::RObject result;
  Rcpp::RNGScope rng_scope;
  // run() takes its three sizes by reference (src/run.rcpp.cpp:58): named objects, not temporaries
  arg<unsigned &> N_(N), d_(d), T_(timeSteps);
  result = Rcpp::wrap(run(N_, d_, T_, arg<Mat>(Y), arg<Vec>(m0), arg<Mat>(C0), arg<Mat>(F), arg<Mat>(G), arg<Mat>(V),
                          arg<Mat>(W), arg<float>(df), arg<std::string>(resampler), arg<std::string>(distribution),
                          arg<unsigned>(p)));
  return result;
  END_RCPP
}

RcppExport SEXP _CuSMC_metropolis_hastings(SEXP w, SEXP N, SEXP B)
{
  CUSMC_SHIM(metropolis_hastings(arg<Vec>(w), arg<int>(N), arg<int>(B)))
}

#undef CUSMC_SHIM

// ---- registration: same names and arity as the reference's table --------------------------------------
#define CUSMC_ENTRY(name, arity) {#name, (DL_FUNC)&name, arity}
static const R_CallMethodDef kCallEntries[] = {
    CUSMC_ENTRY(_CuSMC_MVN, 2),
    CUSMC_ENTRY(_CuSMC_MVNPDF, 3),
    CUSMC_ENTRY(_CuSMC_MVT, 3),
    CUSMC_ENTRY(_CuSMC_MVTPDF, 4),
    CUSMC_ENTRY(_CuSMC_run, 14),
    CUSMC_ENTRY(_CuSMC_metropolis_hastings, 3),
    {NULL, NULL, 0}};
#undef CUSMC_ENTRY

RcppExport void R_init_CuSMC(DllInfo *dll)
{
  R_registerRoutines(dll, NULL, kCallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

// library.dynam.unload("CuSMC", ...): release the cached distributions, then the context, while the HIP
// runtime is certainly still alive.  At process exit nothing is released by this library (the OS does it).
RcppExport void R_unload_CuSMC(DllInfo *)
{
  delete cusmc_glue::g_session;
  cusmc_glue::g_session = nullptr;
}
