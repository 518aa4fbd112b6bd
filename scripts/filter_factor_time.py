"""The filter loop (cusmc_pf_run_host, outputs not copied out) with dense G and W: eigenSolver's square roots (the
default, as MCMC() computes them) against CUSMC_PROPOSAL_FACTOR=cholesky (triangular proposal kernels from d = 32 up).
    python scripts/filter_factor_time.py [tag]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402
from cusmc_amd import _lib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
L = _lib.lib()
ctx = cusmc_amd.api.default_context()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def spd(rng, d):
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


def run(N, d, T, dist, nu, reps=3):
    rng = np.random.default_rng(1)
    Y = np.ascontiguousarray(np.cumsum(0.1 * rng.standard_normal((T, d)), axis=0))
    G = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d / 8)
    m0, C0, F, V, W = np.zeros(d), spd(rng, d), np.eye(d), spd(rng, d), 0.3 * spd(rng, d)
    tail = (ptr(Y), N, d, T, ptr(m0), ptr(C0), ptr(F), ptr(G), ptr(V), ptr(W), C.c_float(nu), b"metropolis", dist, 10,
            C.c_double(1.0), 7, None, None, None)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = L.cusmc_pf_run_host(ctx._h, *tail)
        assert rc == 0, L.cusmc_last_error()
        best = min(best, time.perf_counter() - t0)
    return best


for N, d, T in ((200_000, 64, 20), (200_000, 112, 10), (200_000, 128, 10), (100_000, 256, 10)):
    for dist, nu in ((b"mvn", 0.0), (b"mvt", 4.0)):
        t = {}
        for factor in ("eigen", "cholesky"):
            os.environ["CUSMC_PROPOSAL_FACTOR"] = factor
            t[factor] = run(N, d, T, dist, nu)
        print("%s N=%d d=%d T=%d %s: eigen %.2f ms (%.1f us per step) | cholesky %.2f ms (%.1f us per step)" % (
            tag, N, d, T, dist.decode(), t["eigen"] * 1e3, t["eigen"] * 1e6 / (T - 1), t["cholesky"] * 1e3,
            t["cholesky"] * 1e6 / (T - 1)), flush=True)
