// Small dense fp64 linear algebra on the host, row-major.  Everything here runs once per
// distribution object (O(d^3), d <= 256): it replaces the Eigen calls the reference makes per
// PARTICLE -- sigma.determinant(), sigma.inverse() (src/statistics.cc.cpp:176-177,190-193,
// 301,306) -- and the eigenSolver of src/linear_algebra.cpp:10-23.  No Eigen in this library.
#pragma once
#include <algorithm>
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

inline bool is_lower_triangular(const double *A, int n)
{
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (A[(size_t)i * n + j] != 0.0) return false;
  return true;
}

// QL factorisation by Householder reflections: M = Q L with Q orthogonal and L lower triangular
// (both n x n, row-major).  Returns L and Qt = Q^T, so that for any b
//     |b + M x|^2 = |Qt b + L x|^2.
// Used to give reweight_G's dense matrix -W F (src/mcmc.cpp:208 via statistics.cc.cpp:171-180) the
// triangular shape the matrix-core kernels run; works for singular M as well.
inline void ql_factor(const double *M, int n, std::vector<double> &L, std::vector<double> &Qt)
{
  L.assign(M, M + (size_t)n * n);
  Qt.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Qt[(size_t)i * n + i] = 1.0;
  std::vector<double> v(n);
  for (int k = n - 1; k >= 1; --k) {
    // reflect rows 0..k so that column k becomes (0, ..., 0, alpha)
    double scale = 0.0;
    for (int i = 0; i <= k; ++i) scale = std::fmax(scale, std::fabs(L[(size_t)i * n + k]));
    if (scale == 0.0) continue;
    double norm2 = 0.0;
    for (int i = 0; i <= k; ++i) {
      v[i] = L[(size_t)i * n + k] / scale;
      norm2 += v[i] * v[i];
    }
    const double norm = std::sqrt(norm2);
    const double alpha = v[k] > 0.0 ? -norm : norm;  // (no cancellation in v[k] - alpha)
    const double vk = v[k] - alpha;
    const double vnorm2 = norm2 - v[k] * v[k] + vk * vk;
    v[k] = vk;
    if (vnorm2 == 0.0) continue;
    const double beta = 2.0 / vnorm2;
    for (int c = 0; c < k; ++c) {  // H = I - beta v v^T on the columns still to do
      double s = 0.0;
      for (int i = 0; i <= k; ++i) s += v[i] * L[(size_t)i * n + c];
      s *= beta;
      for (int i = 0; i <= k; ++i) L[(size_t)i * n + c] -= s * v[i];
    }
    for (int i = 0; i < k; ++i) L[(size_t)i * n + k] = 0.0;
    L[(size_t)k * n + k] = alpha * scale;
    for (int c = 0; c < n; ++c) {
      double s = 0.0;
      for (int i = 0; i <= k; ++i) s += v[i] * Qt[(size_t)i * n + c];
      s *= beta;
      for (int i = 0; i <= k; ++i) Qt[(size_t)i * n + c] -= s * v[i];
    }
  }
}

// Q = V sqrt(Lambda) from the symmetric eigen-decomposition S = V Lambda V^T, the matrix
// eigenSolver() builds (src/linear_algebra.cpp:13-22 through Eigen's SelfAdjointEigenSolver).
// Same route as Eigen's: Householder reduction to tridiagonal form, then implicit QL sweeps (the
// EISPACK tred2 / tql2 pair), O(n^3) with a small constant -- the cyclic Jacobi this replaces took
// 0.9 s at n = 256.  Eigenvalues ascending, as Eigen returns them; each eigenvector's sign fixed
// so that its largest component is positive (Eigen leaves it to the algorithm), which makes Q
// reproducible across implementations -- the oracle does the same.  Q Q^T = S; negative round-off
// eigenvalues are clamped to zero.
inline void eigen_sqrt(const double *S, int n, double *Q)
{
  std::vector<double> V(S, S + (size_t)n * n), d(n), e(n);
  // column-major working storage: every O(n^3) loop below walks down a column
  auto v = [&](int i, int j) -> double & { return V[(size_t)j * n + i]; };
  // ---- tred2: V <- orthogonal transformation to tridiagonal form; d = diagonal, e = sub-diagonal
  for (int j = 0; j < n; ++j) d[j] = v(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
        v(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1], g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        v(j, i) = f;
        g = e[j] + v(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) {
          g += v(k, j) * d[k];
          e[k] += v(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) v(k, j) -= (f * e[k] + g * d[k]);
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    v(n - 1, i) = v(i, i);
    v(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = v(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += v(k, i + 1) * v(k, j);
        for (int k = 0; k <= i; ++k) v(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) v(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) {
    d[j] = v(n - 1, j);
    v(n - 1, j) = 0.0;
  }
  v(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- tql2: implicit QL on the tridiagonal matrix, rotations accumulated into V
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m > l) {
      for (int iter = 0; iter < 200; ++iter) {
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = v(k, i + 1);
            v(k, i + 1) = s * v(k, i) + c * h;
            v(k, i) = c * v(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
        if (std::fabs(e[l]) <= eps * tst1) break;
      }
    }
    d[l] += f;
    e[l] = 0.0;
  }
  // ascending eigenvalues (selection sort, stable for ties), then the sign convention
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j)
      if (d[j] < p) {
        k = j;
        p = d[j];
      }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int r = 0; r < n; ++r) std::swap(v(r, i), v(r, k));
    }
  }
  for (int j = 0; j < n; ++j) {
    int im = 0;
    for (int i = 1; i < n; ++i)
      if (std::fabs(v(i, j)) > std::fabs(v(im, j))) im = i;
    const double sg = v(im, j) < 0.0 ? -1.0 : 1.0;
    const double root = d[j] > 0.0 ? std::sqrt(d[j]) : 0.0;
    for (int i = 0; i < n; ++i) Q[(size_t)i * n + j] = sg * v(i, j) * root;
  }
}

}  // namespace la
}  // namespace cusmc
