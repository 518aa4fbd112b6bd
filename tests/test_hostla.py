"""Host-side linear algebra of the library (cusmc_amd/csrc/hostla.h) against numpy, on the CPU.

ql_factor() is what lets reweight_G's dense matrix -W F (src/mcmc.cpp:208, pdf of
src/statistics.cc.cpp:171-180) run as a triangular product on the matrix cores: |b + M x|^2 =
|Q^T b + L x|^2 for M = Q L."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hostla") / "hostla_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "host", "hostla_shim.cpp")])
    lib = ctypes.CDLL(so)
    lib.shim_ql_factor.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.shim_ql_factor.restype = None
    return lib


def ql(shim, M):
    n = M.shape[0]
    M = np.ascontiguousarray(M, dtype=np.float64)
    L, Qt = np.empty((n, n)), np.empty((n, n))
    shim.shim_ql_factor(M.ctypes.data, n, L.ctypes.data, Qt.ctypes.data)
    return L, Qt


@pytest.mark.parametrize("n", [1, 2, 3, 16, 17, 64, 200])
def test_ql_factor_reconstructs(shim, n):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n)) * np.exp(rng.uniform(-3, 3, size=(n, 1)))
    L, Qt = ql(shim, M)
    assert np.all(np.triu(L, 1) == 0.0)
    scale = np.abs(M).max()
    assert np.abs(Qt.T @ L - M).max() <= 1e-13 * scale * n
    assert np.abs(Qt @ Qt.T - np.eye(n)).max() <= 1e-13 * n
    # the identity the kernels rely on
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    want = np.sum((b + M @ x) ** 2)
    got = np.sum((Qt @ b + L @ x) ** 2)
    assert abs(got - want) <= 1e-12 * want


def test_ql_factor_singular_and_structured(shim):
    rng = np.random.default_rng(0)
    n = 24
    A = rng.standard_normal((n, 5))
    for M in (A @ A.T, np.zeros((n, n)), np.eye(n), np.tril(rng.standard_normal((n, n))),
              np.triu(rng.standard_normal((n, n)))):
        L, Qt = ql(shim, M)
        assert np.all(np.isfinite(L)) and np.all(np.isfinite(Qt))
        assert np.all(np.triu(L, 1) == 0.0)
        assert np.abs(Qt.T @ L - M).max() <= 1e-12 * max(1.0, np.abs(M).max()) * n
        assert np.abs(Qt @ Qt.T - np.eye(n)).max() <= 1e-13 * n
