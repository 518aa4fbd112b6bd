// Small dense fp64 linear algebra on the host, row-major.  Everything here runs once per
// distribution object (O(d^3), d <= 256): it replaces the Eigen calls the reference makes per
// PARTICLE -- sigma.determinant(), sigma.inverse() (src/statistics.cc.cpp:176-177,190-193,
// 301,306) -- and the eigenSolver of src/linear_algebra.cpp:10-23.  No Eigen in this library.
#pragma once
#include <cmath>
#include <vector>

namespace cusmc {
namespace la {

// Sigma = L L^T, lower triangular L (row-major, upper part zeroed).  Returns 0, or 1 + the
// index of the first non-positive pivot; rejects a visibly asymmetric input with -1.
inline int cholesky(const double *S, int n, std::vector<double> &L)
{
  L.assign((size_t)n * n, 0.0);
  double scale = 0.0;
  for (int i = 0; i < n * n; ++i) scale = std::fmax(scale, std::fabs(S[i]));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (std::fabs(S[i * n + j] - S[j * n + i]) > 1e-9 * scale) return -1;
  for (int j = 0; j < n; ++j) {
    double diag = S[j * n + j];
    for (int k = 0; k < j; ++k) diag -= L[j * n + k] * L[j * n + k];
    if (!(diag > 0.0) || !std::isfinite(diag)) return j + 1;
    const double ljj = std::sqrt(diag);
    L[j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = 0.5 * (S[i * n + j] + S[j * n + i]);
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / ljj;
    }
  }
  return 0;
}

// W = L^-1 for lower-triangular L (forward substitution, column by column).
inline void lower_inverse(const std::vector<double> &L, int n, std::vector<double> &W)
{
  W.assign((size_t)n * n, 0.0);
  for (int c = 0; c < n; ++c) {
    W[c * n + c] = 1.0 / L[c * n + c];
    for (int i = c + 1; i < n; ++i) {
      double s = 0.0;
      for (int k = c; k < i; ++k) s += L[i * n + k] * W[k * n + c];
      W[i * n + c] = -s / L[i * n + i];
    }
  }
}

// C = A B (n x n).
inline void matmul(const double *A, const double *B, int n, std::vector<double> &C)
{
  C.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) {
      const double a = A[i * n + k];
      if (a == 0.0) continue;
      for (int j = 0; j < n; ++j) C[i * n + j] += a * B[k * n + j];
    }
}

inline void matvec(const double *A, const double *x, int n, std::vector<double> &y)
{
  y.assign(n, 0.0);
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += A[i * n + j] * x[j];
    y[i] = s;
  }
}

inline bool is_identity(const double *F, int n)
{
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      if (F[i * n + j] != (i == j ? 1.0 : 0.0)) return false;
  return true;
}

inline bool is_diagonal(const double *A, int n)
{
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      if (i != j && A[i * n + j] != 0.0) return false;
  return true;
}

// Q = V sqrt(Lambda) from the symmetric eigen-decomposition (cyclic Jacobi), the matrix
// eigenSolver() builds (src/linear_algebra.cpp:13-22).  Q Q^T = S; negative round-off
// eigenvalues are clamped to zero.
inline void eigen_sqrt(const double *S, int n, double *Q)
{
  std::vector<double> A(S, S + (size_t)n * n), V((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += 2.0 * A[i * n + j] * A[i * n + j];
    }
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double tau = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = std::copysign(1.0, tau) / (std::fabs(tau) + std::hypot(1.0, tau));
        const double c = 1.0 / std::hypot(1.0, t), s = t * c;
        for (int k = 0; k < n; ++k) {  // columns p, q of A and V
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
        for (int k = 0; k < n; ++k) {  // rows p, q of A
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const double lam = A[j * n + j];
      Q[i * n + j] = V[i * n + j] * (lam > 0.0 ? std::sqrt(lam) : 0.0);
    }
}

}  // namespace la
}  // namespace cusmc
