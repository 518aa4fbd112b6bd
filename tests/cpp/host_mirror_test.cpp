// Drives cusmc_amd/host/cusmc_host.hpp (the C++ mirror of the reference's classes) on a GPU box.
// Built by __graft_entry__.build() with g++ (no HIP needed: it only links the C ABI) and run by
// tests/test_gpu_parity.py::test_cpp_host_mirror.  Checks the reference's published values.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../cusmc_amd/host/cusmc_host.hpp"

using namespace cusmc;

static int fails = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++fails; } \
  } while (0)

int main()
{
  try {
    {  // CuSMC/CuSMC.tex:95-105: MVNPDF(c(0,0), c(0,0), diag(2)) = 0.1591549
      MultiVariateNormalDistribution MVN(Vector{0, 0}, Matrix::Identity(2));
      const double p = MVN.pdf(Vector{0, 0}, Matrix::Identity(2));
      EXPECT(std::fabs(p - 0.15915494309189535) < 1e-15);
      EXPECT(std::fabs(MVN.getNorm() - 0.15915494309189535) < 1e-15);
      EXPECT(std::fabs(MVN.pdf(Vector{0, 0}) - p) < 1e-16);  // the one-argument overload
    }
    {  // CuSMC/CuSMC.tex:131-142: MVTPDF(c(0,0,0), c(0,0,0), diag(3), 3.0) = 0.07799708
      MultiVariateTStudentDistribution MVT(Vector{0, 0, 0}, Matrix::Identity(3), 3.0f);
      EXPECT(std::fabs(MVT.pdf(Vector{0, 0, 0}, Matrix::Identity(3)) - 0.0779970835340203) < 1e-15);
      EXPECT(MVT.dfree() == 3.0f);
    }
    {  // man/metropolis_hastings.Rd:22-27: w = c(0,0), N = 2, B = 10 -> c(0, 1)
      unsigned a[4] = {9, 9, 9, 9};
      Vector w[1] = {Vector{0.0, 0.0}};
      Sampler::metropolis_hastings(a, w, 2, 1, 10);
      EXPECT(a[2] == 0 && a[3] == 1 && a[0] == 9);  // writes a_t[t*N + i] only
    }
    {  // pdf(y, F) subtracts F mu; batched == scalar; non-identity F
      Matrix S(2, 2); S(0, 0) = 2; S(1, 1) = 1; S(0, 1) = S(1, 0) = 0.5;
      Matrix F(2, 2); F(0, 0) = 1; F(0, 1) = 2; F(1, 0) = 0; F(1, 1) = 1;   // asymmetric on purpose
      MultiVariateNormalDistribution D(Vector{1, -1}, S);
      const double X[4] = {0.3, 0.2, -1.0, -1.0};
      double out[2];
      D.pdf(X, 2, &F, out);
      EXPECT(std::fabs(out[0] - D.pdf(Vector{0.3, 0.2}, F)) < 1e-16);
      // closed form: r = x - F mu, F mu = (1 - 2, -1) = (-1, -1)
      const double r0 = 0.3 + 1, r1 = 0.2 + 1, det = 2 * 1 - 0.25;
      const double q = (1 * r0 * r0 - 2 * 0.5 * r0 * r1 + 2 * r1 * r1) / det;
      EXPECT(std::fabs(out[0] - std::exp(-0.5 * q) / (2 * M_PI * std::sqrt(det))) < 1e-15);
      EXPECT(std::fabs(out[1] - 1.0 / (2 * M_PI * std::sqrt(det))) < 1e-15);  // x = F mu
    }
    {  // errors surface as exceptions with the ABI's message (Rcpp::stop in the R build)
      bool threw = false;
      Matrix bad(2, 2); bad(0, 0) = 1; bad(1, 1) = 1; bad(0, 1) = bad(1, 0) = 2;
      try { MultiVariateNormalDistribution D(Vector{0, 0}, bad); } catch (const Error &e) { threw = e.code == CUSMC_ENOTSPD; }
      EXPECT(threw);
    }
    {  // eigenSolver: Q Q^T = sigma
      Matrix S(2, 2); S(0, 0) = 2; S(1, 1) = 1; S(0, 1) = S(1, 0) = 0.5;
      Matrix Q; eigenSolver(Q, S);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) EXPECT(std::fabs(Q(i, 0) * Q(j, 0) + Q(i, 1) * Q(j, 1) - S(i, j)) < 1e-13);
    }
    {  // particle_filter with the declaration's argument order; unknown options are rejected
      const unsigned N = 128, d = 2, T = 4;
      std::vector<double> X(T * N * d), w(T * N), y(T * d, 0.1);
      std::vector<unsigned> a(T * N);
      const Matrix I = Matrix::Identity(2);
      particle_filter(X.data(), w.data(), a.data(), y.data(), I, I, Vector{0, 0}, I, I, I, N, d, T, 0.f, "metropolis", "mvn", 7);
      EXPECT(std::fabs(w[0] - 1.0 / N) < 1e-18);
      for (unsigned i = 0; i < N; ++i) EXPECT(a[N + i] < N && w[N + i] > 0.0);
      bool threw = false;
      try { particle_filter(X.data(), w.data(), a.data(), y.data(), I, I, Vector{0, 0}, I, I, I, N, d, T, 0.f, "systematic", "mvn"); }
      catch (const Error &e) { threw = e.code == CUSMC_EINVAL; }
      EXPECT(threw);
      // three shards rehearsed on device 0: the sharded loop returns the same history bit for bit
      std::vector<double> X2(T * N * d), w2(T * N);
      std::vector<unsigned> a2(T * N);
      particle_filter_multi({0, 0, 0}, X2.data(), w2.data(), a2.data(), y.data(), I, I, Vector{0, 0}, I, I, I, N, d, T, 0.f,
                            "metropolis", "mvn", 7);
      EXPECT(X2 == X && w2 == w && a2 == a);
      EXPECT(stream_key(0, 0) == 0xE220A8397B1DCDAFull && stream_key(1, 0) != stream_key(1, 1));
    }
  } catch (const std::exception &e) {
    std::printf("FAIL exception: %s\n", e.what());
    return 2;
  }
  std::printf(fails ? "host mirror: %d failure(s)\n" : "host mirror: all checks passed\n", fails);
  return fails ? 1 : 0;
}
