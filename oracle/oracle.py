"""ctypes binding of oracle/libcusmc_oracle.so (the CPU restatement, see cusmc_oracle.c).

TEST INFRASTRUCTURE.  Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under cusmc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcusmc_oracle.so")
_lib = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "cusmc_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_det.restype = C.c_double
        L.oracle_mvn_norm.restype = C.c_double
        L.oracle_mvt_norm.restype = C.c_double
        L.oracle_mvn_pdf.restype = C.c_double
        L.oracle_mvt_pdf.restype = C.c_double
        L.oracle_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _opt(a):
    return None if a is None else _d(a).ctypes.data_as(C.c_void_p)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def det(S):
    S = _d(S)
    return lib().oracle_det(_p(S), C.c_int(S.shape[0]))


def inverse(S):
    S = _d(S)
    out = np.empty_like(S)
    lib().oracle_inverse(_p(S), _p(out), C.c_int(S.shape[0]))
    return out


def mvn_norm(sigma):
    sigma = _d(sigma)
    return lib().oracle_mvn_norm(_p(sigma), C.c_int(sigma.shape[0]))


def mvt_norm(sigma, nu):
    sigma = _d(sigma)
    return lib().oracle_mvt_norm(_p(sigma), C.c_int(sigma.shape[0]), C.c_float(nu))


def mvn_pdf(y, mu, sigma, F=None):
    """MultiVariateNormalDistribution::pdf(y[,F]) restated (src/statistics.cc.cpp:171-196)."""
    y, sigma = _d(y), _d(sigma)
    mu = _d(np.zeros_like(y) if mu is None else mu)
    Fk = None if F is None else _d(F)
    return lib().oracle_mvn_pdf(_p(y), _p(mu), _p(sigma), None if Fk is None else _p(Fk),
                                C.c_int(y.shape[0]))


def mvt_pdf(y, mu, sigma, nu, F=None):
    """MultiVariateTStudentDistribution::pdf(y[,F]) restated (src/statistics.cc.cpp:295-324)."""
    y, sigma = _d(y), _d(sigma)
    mu = _d(np.zeros_like(y) if mu is None else mu)
    Fk = None if F is None else _d(F)
    return lib().oracle_mvt_pdf(_p(y), _p(mu), _p(sigma), None if Fk is None else _p(Fk),
                                C.c_int(y.shape[0]), C.c_float(nu))


def pdf_batch(X, mu, sigma, F=None, dist="mvn", nu=0.0):
    """Reference-faithful batched density: per-particle LU det + inverse."""
    X, sigma = _d(X), _d(sigma)
    N, d = X.shape
    mu = _d(np.zeros(d) if mu is None else mu)
    Fk = None if F is None else _d(F)
    out = np.empty(N)
    lib().oracle_pdf_batch(_p(X), C.c_long(N), C.c_long(d), _p(mu), _p(sigma),
                           None if Fk is None else _p(Fk), C.c_int(d),
                           C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), _p(out))
    return out


def reweight(X, y, F, V, dist="mvn", nu=0.0):
    """reweight_G CPU branch (src/mcmc.cpp:185-215): w_i = pdf_{0,V}(y - F x_i)."""
    X, y, F, V = _d(X), _d(y), _d(F), _d(V)
    N, d = X.shape
    w = np.empty(N)
    lib().oracle_reweight(_p(X), C.c_long(N), C.c_long(d), _p(y), _p(F), _p(V), C.c_int(d),
                          C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), _p(w))
    return w


def logpdf_hoisted(X, mu, sigma, F=None, dist="mvn", nu=0.0):
    X, sigma = _d(X), _d(sigma)
    N, d = X.shape
    mu = _d(np.zeros(d) if mu is None else mu)
    Fk = None if F is None else _d(F)
    out = np.empty(N)
    rc = lib().oracle_logpdf_hoisted(_p(X), C.c_long(N), C.c_long(d), _p(mu), _p(sigma),
                                     None if Fk is None else _p(Fk), C.c_int(d),
                                     C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), _p(out))
    if rc:
        raise ValueError("sigma is not symmetric positive definite (pivot %d)" % rc)
    return out


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def metropolis(w, B, seed, step=1, N=None, first=0, count=None):
    """Sampler::metropolis_hastings inner loop (src/samplers.cpp:21-35) under the build's
    Philox contract.  Returns 0-based ancestors (uint32) of chains [first, first + count)
    (default: all N)."""
    w = _d(w)
    N = w.shape[0] if N is None else N
    count = N - first if count is None else count
    a = np.empty(count, dtype=np.uint32)
    lib().oracle_metropolis_range(_p(a), _p(w), C.c_uint32(N), C.c_uint32(B), C.c_uint64(seed),
                                  C.c_uint32(step), C.c_uint32(first), C.c_uint32(count))
    return a


def metropolis_log(lw, B, seed, step=1):
    """The same chain over log-weights: accept iff u <= exp(lw[j] - lw[k])."""
    lw = _d(lw)
    N = lw.shape[0]
    a = np.empty(N, dtype=np.uint32)
    lib().oracle_metropolis_log(_p(a), _p(lw), C.c_uint32(N), C.c_uint32(B), C.c_uint64(seed), C.c_uint32(step))
    return a


def exp_nonpos(t):
    f = lib().oracle_exp_nonpos
    f.restype = C.c_double
    return f(C.c_double(t))


def pdf_percov(X, mu, sigma, dist="mvn", nu=0.0):
    """The reference's pdf() with a distribution object per particle (densities)."""
    X, sigma = _d(X), _d(sigma)
    N, d = X.shape
    ldmu = 0
    if mu is not None:
        mu = _d(mu)
        ldmu = d if mu.ndim == 2 else 0
    out = np.empty(N)
    lib().oracle_pdf_percov(_p(X), C.c_long(N), C.c_long(d), None if mu is None else _p(mu), C.c_long(ldmu),
                            _p(sigma), C.c_int(d), C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), _p(out))
    return out


def chol_batched(sigma):
    """Batched Cholesky in the kernels' operation order: (L, logdet, info)."""
    sigma = _d(sigma)
    N, d = sigma.shape[0], sigma.shape[1]
    L, logdet, info = np.empty_like(sigma), np.empty(N), np.empty(N, dtype=np.int32)
    lib().oracle_chol_batched(_p(sigma), C.c_long(N), C.c_int(d), _p(L), _p(logdet), _p(info))
    return L, logdet, info


def chi_square(count, d, nu, seed=0, step=0, first=0):
    """The chi^2_nu draws of components 0..d-1 of particles first..first+count-1 (RNG contract 2)."""
    out = np.empty((count, d), dtype=np.float64)
    lib().oracle_chi_square(_p(out), C.c_uint32(first), C.c_uint32(count), C.c_int(d), C.c_float(nu),
                            C.c_uint64(seed), C.c_uint32(step))
    return out


def rng_contract():
    return int(lib().oracle_rng_contract())


def eigen_sqrt(S):
    S = _d(S)
    Q = np.empty_like(S)
    lib().oracle_eigen_sqrt(_p(S), _p(Q), C.c_int(S.shape[0]))
    return Q


def initialize(N, m0, Q0, dist="mvn", nu=0.0, scale=1.0, seed=0, step=0):
    m0, Q0 = _d(m0), _d(Q0)
    d = m0.shape[0]
    X0 = np.empty((N, d))
    w0 = np.empty(N)
    lib().oracle_initialize(_p(X0), _p(w0), C.c_uint32(N), C.c_int(d), _p(m0), _p(Q0),
                            C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), C.c_double(scale),
                            C.c_uint64(seed), C.c_uint32(step))
    return X0, w0


def propagate(Xprev, a, G, Qw, dist="mvn", nu=0.0, scale=1.0, seed=0, step=1):
    Xprev, G, Qw = _d(Xprev), _d(G), _d(Qw)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    N, d = Xprev.shape
    Xt = np.empty((N, d))
    lib().oracle_propagate(_p(Xt), _p(Xprev), _p(a), C.c_uint32(N), C.c_int(d), _p(G), _p(Qw),
                           C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), C.c_double(scale),
                           C.c_uint64(seed), C.c_uint32(step))
    return Xt


def pf_run(Y, N, m0, C0, F, G, V, W, dist="mvn", nu=0.0, B=10, scale=1.0, seed=0, hoisted=False):
    """particle_filter(): initialize + MCMC loop (src/particle_filter.cpp:22-36,
    src/mcmc.cpp:292-308).  Y is T x d (row t = y_t).  Returns X (T,N,d), w (T,N), a (T,N)."""
    Y, m0, C0, F, G, V, W = map(_d, (Y, m0, C0, F, G, V, W))
    T, d = Y.shape
    X = np.zeros((T, N, d))
    w = np.zeros((T, N))
    a = np.zeros((T, N), dtype=np.uint32)
    rc = lib().oracle_pf_run(_p(X), _p(w), _p(a), _p(Y), C.c_uint32(N), C.c_int(d), C.c_uint32(T),
                             _p(m0), _p(C0), _p(F), _p(G), _p(V), _p(W),
                             C.c_int(0 if dist == "mvn" else 1), C.c_float(nu), C.c_uint32(B),
                             C.c_double(scale), C.c_uint64(seed), C.c_int(1 if hoisted else 0))
    if rc:
        raise ValueError("V is not symmetric positive definite")
    return X, w, a


def num_threads():
    return lib().oracle_num_threads()
