# R stubs of the six exports -- what Rcpp::compileAttributes() writes from the roxygen blocks in
# src/*.rcpp.cpp, kept in the tree so that the package installs without that step.  Same function
# names, argument lists and .Call symbols as the reference's R/RcppExports.R:17-103.

#' @useDynLib CuSMC
#' @importFrom Rcpp evalCpp
NULL

#' MultiVariateNormal Distribution
#'
#' @param mu     [vector]: Mean vector.
#' @param sigma  [matrix]: Covariance matrix.
#' @return       [vector]: one draw from N(mu, sigma); options(CuSMC.compat = TRUE) reproduces the
#'               reference's transform (sigma itself as the factor, variance inflated 3x).
#' @export
#' @examples
#' CuSMC::MVN(c(0, 0), diag(2))
MVN <- function(mu, sigma) {
    .Call('_CuSMC_MVN', PACKAGE = 'CuSMC', mu, sigma)
}

#' MultiVariateNormal Probability Density Function
#'
#' @param x      [vector | d x N matrix]: one point, or particles in columns.
#' @param mu     [vector]: Mean vector.
#' @param sigma  [matrix]: Covariance matrix.
#' @return       [numeric]: the density at x (one number), or one density per column of x.
#' @export
#' @examples
#' CuSMC::MVNPDF(c(0, 0), c(0, 0), diag(2))            # 0.1591549
#' CuSMC::MVNPDF(matrix(rnorm(2 * 1e4), 2), c(0, 0), diag(2))
MVNPDF <- function(x, mu, sigma) {
    .Call('_CuSMC_MVNPDF', PACKAGE = 'CuSMC', x, mu, sigma)
}

#' MultiVariate T Distribution
#'
#' @param mu     [vector]: Location vector.
#' @param sigma  [matrix]: Dispersion matrix.
#' @param nu     [float]: degrees of freedom.
#' @return       [vector]: one draw.
#' @export
MVT <- function(mu, sigma, nu) {
    .Call('_CuSMC_MVT', PACKAGE = 'CuSMC', mu, sigma, nu)
}

#' MultiVariate T Probability Density Function
#'
#' @param x      [vector | d x N matrix]: one point, or particles in columns.
#' @param mu     [vector]: Location vector.
#' @param sigma  [matrix]: Dispersion matrix.
#' @param nu     [float]: degrees of freedom.
#' @return       [numeric]: the density at x (one number), or one density per column of x.
#' @export
#' @examples
#' CuSMC::MVTPDF(c(0, 0, 0), c(0, 0, 0), diag(3), 3.0)  # 0.07799708
MVTPDF <- function(x, mu, sigma, nu) {
    .Call('_CuSMC_MVTPDF', PACKAGE = 'CuSMC', x, mu, sigma, nu)
}

#' Run the bootstrap particle filter
#'
#' @param N            [integer]: Number of particles.
#' @param d            [integer]: State dimension.
#' @param timeSteps    [integer]: Number of time steps.
#' @param Y            [matrix]: Observations, d x timeSteps (columns = time).
#' @param m0           [vector]: Initial mean.
#' @param C0           [matrix]: Initial covariance.
#' @param F            [matrix]: Observation matrix.
#' @param G            [matrix]: Transition matrix.
#' @param V            [matrix]: Observation noise covariance.
#' @param W            [matrix]: State noise covariance.
#' @param df           [float]: Degrees of freedom (distribution = "mvt").
#' @param resampler    [string]: "metropolis".
#' @param distribution [string]: "mvn" or "mvt".
#' @param p            [integer]: particle written to x_t_N<p>.csv.
#' @return list(weights = timeSteps x N matrix, posterior_x = timeSteps x N x d array)
#' @export
run <- function(N, d, timeSteps, Y, m0, C0, F, G, V, W, df, resampler, distribution, p = 0L) {
    .Call('_CuSMC_run', PACKAGE = 'CuSMC', N, d, timeSteps, Y, m0, C0, F, G, V, W, df, resampler, distribution, p)
}

#' Metropolis Hastings Sampler
#'
#' @param w  [vector]: Weights.
#' @param N  [integer]: Number of weights.
#' @param B  [integer]: Accept/reject iterations per particle.
#' @return   [vector]: Ancestors (0-based, as doubles).
#' @export
#' @examples
#' CuSMC::metropolis_hastings(c(0, 0), 2, 10)            # 0 1
metropolis_hastings <- function(w, N, B) {
    .Call('_CuSMC_metropolis_hastings', PACKAGE = 'CuSMC', w, N, B)
}
