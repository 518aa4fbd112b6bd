// Test shim: exposes the library's host linear algebra (cusmc_amd/csrc/hostla.h, header-only,
// no HIP) to ctypes so that the CPU suite can check it against numpy.
#include "../../cusmc_amd/csrc/hostla.h"

extern "C" void shim_ql_factor(const double *M, int n, double *L, double *Qt)
{
  std::vector<double> l, q;
  cusmc::la::ql_factor(M, n, l, q);
  std::copy(l.begin(), l.end(), L);
  std::copy(q.begin(), q.end(), Qt);
}
