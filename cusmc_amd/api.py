"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Same names and argument meaning as the reference:

  R level  (R/RcppExports.R:17-103; bodies src/*.rcpp.cpp)
      MVN(mu, sigma)  MVNPDF(x, mu, sigma)  MVT(mu, sigma, nu)  MVTPDF(x, mu, sigma, nu)
      metropolis_hastings(w, N, B)  run(N, d, timeSteps, Y, m0, C0, F, G, V, W, df, resampler,
      distribution, p=0)
  C++ level (inst/include/statistics.hpp:36-250, inst/include/samplers.hpp:7-18)
      MultiVariateNormalDistribution(mu, sigma) / MultiVariateTStudentDistribution(mu, sigma, nu)
      with pdf(y), pdf(y, F), getNorm(), sample(Q, n);  Sampler.metropolis_hastings(...)

This module holds no arithmetic of its own: every number comes out of libcusmc_hip.so.  The
Rcpp glue in rcpp/src binds the same C ABI for R (it cannot be compiled in this image, which
has no R), so this is the layer the parity tests drive.

Conventions carried over from R: matrices arrive as numpy arrays indexed [row, col] like R's;
`Y` is d x T with columns = time (src/run.rcpp.cpp:91); a batched `x` is d x N with columns =
particles.  Densities (not logs) are returned wherever the reference returns densities.
"""
import atexit
import collections
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from ._lib import MVN as _MVN, MVT as _MVT, OUT_DENSITY, OUT_LOG, CusmcError, check

__all__ = ["Context", "MultiVariateNormalDistribution", "MultiVariateTStudentDistribution",
           "Sampler", "propagate_dev", "initialize_dev", "MVN", "MVNPDF", "MVT", "MVTPDF", "metropolis_hastings", "run",
           "set_seed", "eigenSolver", "cholesky_batched", "logpdf_percov", "CusmcError"]

SQRT3 = 1.7320508075688772  # the reference CPU transform's std-dev inflation


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU + stream (cusmc_ctx).  The reference has no such object: it implicitly uses device
    0 and resets it after every call (src/mvn_dist.cu.cpp:788)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        # distribution handles point into their context (device buffers, stream): closing the context
        # closes them first, so that teardown order -- e.g. of module globals at interpreter exit --
        # cannot leave a handle pointing at a destroyed context
        self._distributions = weakref.WeakSet()
        check(_lib.lib().cusmc_ctx_create(device, C.byref(self._h)))

    def set_stream(self, stream_ptr):
        check(_lib.lib().cusmc_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)
        return self

    def synchronize(self):
        check(_lib.lib().cusmc_ctx_synchronize(self._h))

    @property
    def num_cus(self):
        n = C.c_int()
        check(_lib.lib().cusmc_ctx_num_cus(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            for dist in list(self._distributions):
                dist.close()
            _lib.lib().cusmc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None
_rng = {"seed": None, "calls": 0}


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_seed(seed):
    """Seed the R-level draw / resample functions.  The reference cannot be seeded at all
    (std::random_device per call: src/samplers.cpp:10-11); unseeded, this module behaves the same
    way (a fresh seed from the OS), but CUSMC_SEED or set_seed() make runs reproducible."""
    _rng["seed"] = int(seed) & (2 ** 64 - 1)
    _rng["calls"] = 0


def _next_stream():
    if _rng["seed"] is None:
        env = os.environ.get("CUSMC_SEED")
        set_seed(int(env) if env is not None else int.from_bytes(os.urandom(8), "little"))
    _rng["calls"] += 1
    return _rng["seed"], _rng["calls"]


def _next_key():
    """The Philox key of the next R-level call: cusmc_stream_key(session seed, call counter).  Every
    unseeded MVN() / MVT() / metropolis_hastings() / run() gets a key of its own, so repeated calls are
    independent replications (the reference reseeds from std::random_device per call) and no two calls
    share counters, while set_seed() / CUSMC_SEED reproduce the whole sequence of calls."""
    seed, call = _next_stream()
    return int(_lib.lib().cusmc_stream_key(seed, call))


def eigenSolver(sigma):
    """eigenSolver(I_sol, sigma) -- src/linear_algebra.cpp:10-23: Q = V sqrt(Lambda)."""
    sigma = _f64(sigma)
    Q = np.empty_like(sigma)
    check(_lib.lib().cusmc_eigen_sqrt(_ptr(sigma), sigma.shape[0], _ptr(Q)))
    return Q


def cholesky_batched(sigma, ctx=None):
    """Cholesky factors of N small covariances at once (d <= 16).
    sigma: N x d x d.  Returns (L, logdet, info): L N x d x d lower triangular, logdet N, info N
    (0, or 1 + the index of the first non-positive pivot)."""
    ctx = ctx or default_context()
    sigma = _f64(sigma)
    if sigma.ndim != 3 or sigma.shape[1] != sigma.shape[2]:
        raise ValueError("sigma must be N x d x d")
    N, d = sigma.shape[0], sigma.shape[1]
    L, logdet, info = np.empty_like(sigma), np.empty(N), np.empty(N, dtype=np.int32)
    check(_lib.lib().cusmc_chol_batched_host(ctx._h, _ptr(sigma), N, d, _ptr(L), _ptr(logdet), _ptr(info)))
    return L, logdet, info


def logpdf_percov(X, mu, sigma, nu=None, log=True, ctx=None):
    """log p(X[i]; mu[i], sigma[i]) with a covariance PER PARTICLE (d <= 16): multivariate Normal,
    or Student-t when nu is given.  X: N x d; mu: None, a d-vector or N x d; sigma: N x d x d.
    Returns (values, info)."""
    ctx = ctx or default_context()
    X, sigma = _f64(X), _f64(sigma)
    N, d = X.shape
    if sigma.shape != (N, d, d):
        raise ValueError("sigma must be N x d x d")
    ldmu = 0
    if mu is not None:
        mu = _f64(mu)
        if mu.shape == (N, d) and mu.ndim == 2:
            ldmu = d
        elif mu.shape != (d,):
            raise ValueError("mu must be None, a d-vector or N x d")
    out, info = np.empty(N), np.empty(N, dtype=np.int32)
    check(_lib.lib().cusmc_logpdf_percov_host(ctx._h, _MVN if nu is None else _MVT, 0.0 if nu is None else float(nu),
                                              _ptr(X), N, d, None if mu is None else _ptr(mu), ldmu, _ptr(sigma), d,
                                              _lib.OUT_LOG if log else _lib.OUT_DENSITY, _ptr(out), _ptr(info)))
    return out, info


class _Distribution:
    """StatisticalDistribution (inst/include/statistics.hpp:36-96) over a cusmc_dist handle."""
    _kind = _MVN

    def __init__(self, mu, sigma, nu=0.0, ctx=None):
        self.ctx = ctx or default_context()
        self.sigma = _f64(sigma)
        if self.sigma.ndim != 2 or self.sigma.shape[0] != self.sigma.shape[1]:
            raise ValueError("sigma must be a square matrix")
        self.d = self.sigma.shape[0]
        self.mu = _f64(np.zeros(self.d) if mu is None else mu).reshape(-1)
        if self.mu.shape[0] != self.d:
            raise ValueError("mu has %d entries, sigma is %d x %d" % (self.mu.shape[0], self.d, self.d))
        self.nu = float(np.float32(nu))
        self._h = C.c_void_p()
        check(_lib.lib().cusmc_dist_create(self.ctx._h, self._kind, _ptr(self.mu), _ptr(self.sigma),
                                           self.d, C.c_float(self.nu), C.byref(self._h)))
        self.ctx._distributions.add(self)

    # -- scalar interface, as the reference's virtuals ---------------------------------------
    def pdf(self, y, F=None):
        """pdf(y) / pdf(y, F) -- src/statistics.cc.cpp:171-196 (mvn), :295-324 (mvt).
        NB pdf(y) IGNORES mu, exactly like the reference's one-argument overload."""
        y = _f64(y).reshape(1, -1)
        if F is None:
            # pdf(y): quadratic form in y itself -> reweight form with x = 0... i.e. y - I*0
            out = self.reweight(np.zeros((1, self.d)), y.reshape(-1), np.eye(self.d), log=False)
        else:
            out = self.pdf_batch(y, F, log=False)
        return float(out[0])

    def getNorm(self):
        """getNorm() -- src/statistics.cc.cpp:205-211, :332-340 (returned as the reference does,
        not as a log)."""
        return float(np.exp(self.lognorm()))

    def lognorm(self):
        v = C.c_double()
        check(_lib.lib().cusmc_dist_lognorm(self._h, C.byref(v)))
        return v.value

    def logdet(self):
        v = C.c_double()
        check(_lib.lib().cusmc_dist_logdet(self._h, C.byref(v)))
        return v.value

    def mean(self):
        return self.mu.copy()

    def stdev(self):
        return self.sigma.copy()  # sic: the reference returns sigma (statistics.cc.cpp:221)

    # -- batched interface (host buffers) -----------------------------------------------------
    def pdf_batch(self, X, F=None, log=True, devices=None):
        """out[i] = (log) p(X[i]; F mu, Sigma): pdf(y, F) over N x d rows.
        devices = [0, 1, ...]: the rows sharded over those GPUs below the C ABI (cusmc_dist_pdf_multi_host; the
        environment variable CUSMC_DEVICES does the same for callers that cannot pass the list)."""
        X = _f64(X)
        if X.ndim != 2 or X.shape[1] != self.d:
            raise ValueError("X must be N x %d" % self.d)
        Fm = None if F is None else self._square(F, "F")
        out = np.empty(X.shape[0])
        flags = OUT_LOG if log else OUT_DENSITY
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(v) for v in devices])
            check(_lib.lib().cusmc_dist_pdf_multi_host(self._h, devs, len(devices), _ptr(X), X.shape[0], X.shape[1],
                                                       _ptr(Fm), flags, _ptr(out)))
        else:
            check(_lib.lib().cusmc_dist_pdf_host(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(Fm), flags, _ptr(out)))
        return out

    def reweight(self, X, y, F, log=True, devices=None):
        """out[i] = (log) p(y - F X[i]; 0, Sigma): reweight_G (src/mcmc.cpp:185-215).  devices: as pdf_batch."""
        X, y = _f64(X), _f64(y).reshape(-1)
        if X.ndim != 2 or X.shape[1] != self.d or y.shape[0] != self.d:
            raise ValueError("X must be N x %d and y of length %d" % (self.d, self.d))
        Fm = None if F is None else self._square(F, "F")
        out = np.empty(X.shape[0])
        flags = OUT_LOG if log else OUT_DENSITY
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(v) for v in devices])
            check(_lib.lib().cusmc_dist_reweight_multi_host(self._h, devs, len(devices), _ptr(X), X.shape[0], X.shape[1],
                                                            _ptr(y), _ptr(Fm), flags, _ptr(out)))
        else:
            check(_lib.lib().cusmc_dist_reweight_host(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(y), _ptr(Fm), flags,
                                                      _ptr(out)))
        return out

    # -- batched interface (device-resident torch tensors) -----------------------------------
    def _tensor_args(self, X, out):
        import torch
        if not (X.is_cuda and out.is_cuda and X.dtype == torch.float64 and out.dtype == torch.float64):
            raise ValueError("X and out must be float64 CUDA tensors")
        if X.dim() != 2 or X.shape[1] != self.d:
            raise ValueError("X must be N x %d, got %s" % (self.d, tuple(X.shape)))
        if X.stride(1) != 1 or not out.is_contiguous() or out.numel() != X.shape[0]:
            raise ValueError("X must be N x d with unit inner stride; out contiguous of length N")
        ldx = X.stride(0) if X.shape[0] > 1 else X.shape[1]
        if ldx < self.d:
            raise ValueError("rows of X overlap (row stride %d < d = %d)" % (ldx, self.d))
        return X.data_ptr(), X.shape[0], ldx, out.data_ptr()

    def _square(self, M, name):
        M = _f64(M)
        if M.shape != (self.d, self.d):
            raise ValueError("%s must be %d x %d, got %s" % (name, self.d, self.d, M.shape))
        return M

    def pdf_dev(self, X, out, F=None, log=True):
        xp, n, ldx, op = self._tensor_args(X, out)
        Fm = None if F is None else self._square(F, "F")
        check(_lib.lib().cusmc_dist_pdf_dev(self._h, C.c_void_p(xp), n, ldx, _ptr(Fm),
                                            OUT_LOG if log else OUT_DENSITY, C.c_void_p(op)))
        return out

    def reweight_dev(self, X, y, F, out, log=True):
        xp, n, ldx, op = self._tensor_args(X, out)
        y = _f64(y).reshape(-1)
        if y.shape[0] != self.d:
            raise ValueError("y must have %d entries" % self.d)
        Fm = None if F is None else self._square(F, "F")
        check(_lib.lib().cusmc_dist_reweight_dev(self._h, C.c_void_p(xp), n, ldx, _ptr(y), _ptr(Fm),
                                                 OUT_LOG if log else OUT_DENSITY, C.c_void_p(op)))
        return out

    # -- draws ---------------------------------------------------------------------------------
    def sample(self, Q, n_iterations=200, count=1, compat=False, seed=None, step=None):
        """sample(draws, Q, n_iterations) -- src/statistics.cc.cpp:224-259, :355-412.
        `n_iterations` only exists for signature parity: the reference's 200-term sum of
        normals is exactly N(0,3) per component, drawn here in one step when compat=True
        (scale sqrt(3)); the default is the statistically correct N(0,1)."""
        del n_iterations
        if seed is None:
            seed, step = _next_key(), 0
        Q = _f64(Q)
        if Q.shape != (self.d, self.d):
            raise ValueError("Q must be %d x %d" % (self.d, self.d))
        out = np.empty((count, self.d))
        check(_lib.lib().cusmc_sample_host(self.ctx._h, self._kind, C.c_float(self.nu), _ptr(self.mu),
                                           _ptr(Q), self.d, SQRT3 if compat else 1.0, seed,
                                           int(step or 0), count, _ptr(out)))
        return out

    def close(self):
        if self._h:
            _lib.lib().cusmc_dist_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiVariateNormalDistribution(_Distribution):
    """inst/include/statistics.hpp:142-195; src/statistics.cc.cpp:155-272."""
    _kind = _MVN

    def __init__(self, mu, sigma, ctx=None):
        super().__init__(mu, sigma, 0.0, ctx)


class MultiVariateTStudentDistribution(_Distribution):
    """inst/include/statistics.hpp:197-250; src/statistics.cc.cpp:276-424."""
    _kind = _MVT

    def __init__(self, mu, sigma, nu, ctx=None):
        super().__init__(mu, sigma, nu, ctx)

    def dfree(self):
        return self.nu


def _dev_matrix(t, name, d=None):
    """float64 CUDA N x d tensor, rows contiguous and packed (the ABI's proposal / filter entry points take
    no leading dimension: rows are d doubles apart)."""
    import torch
    if not (t.is_cuda and t.dtype == torch.float64 and t.dim() == 2 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float64 CUDA tensor N x d" % name)
    if d is not None and t.shape[1] != d:
        raise ValueError("%s must have %d columns, got %d" % (name, d, t.shape[1]))
    return t


def _dev_vector(t, name, dtypes, n=None):
    if not (t.is_cuda and t.dtype in dtypes and t.is_contiguous()):
        raise ValueError("%s must be a contiguous CUDA tensor of dtype %s" % (name, " / ".join(str(x) for x in dtypes)))
    if n is not None and t.numel() != n:
        raise ValueError("%s must have %d entries, got %d" % (name, n, t.numel()))
    return t


def _index_dtypes():
    import torch
    return (torch.int32, torch.uint32) if hasattr(torch, "uint32") else (torch.int32,)


def _host_square(M, d, name):
    M = _f64(M)
    if M.shape != (d, d):
        raise ValueError("%s must be %d x %d, got %s" % (name, d, d, M.shape))
    return M


class Sampler:
    """inst/include/samplers.hpp:7-18."""

    @staticmethod
    def metropolis_hastings(w, N=None, t=1, B=10, seed=0, ctx=None, devices=None):
        """Sampler::metropolis_hastings(a_t, w_t, N, t, B) -- src/samplers.cpp:7-36.  Takes the
        weight vector w_t[t-1] and returns the row a_t[t*N : (t+1)*N] (uint32, 0-based).
        devices = [0, 1, ...]: the chains sharded over those GPUs below the C ABI (cusmc_metropolis_multi_host;
        CUSMC_DEVICES in the environment does the same), ancestors bit-identical to one device."""
        w = _f64(w).reshape(-1)
        N = w.shape[0] if N is None else int(N)
        if N > w.shape[0]:
            raise ValueError("N = %d exceeds the %d weights given" % (N, w.shape[0]))
        a = np.empty(N, dtype=np.uint32)
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(v) for v in devices])
            check(_lib.lib().cusmc_metropolis_multi_host(devs, len(devices), _ptr(w), N, int(B), int(seed), int(t), 0, _ptr(a)))
            return a
        ctx = ctx or default_context()
        check(_lib.lib().cusmc_metropolis_host(ctx._h, _ptr(w), N, int(B), int(seed), int(t), _ptr(a)))
        return a

    @staticmethod
    def metropolis_hastings_log(logw, N=None, t=1, B=10, seed=None, ctx=None, devices=None):
        """The chain over LOG-weights (accept iff u <= exp(logw[j] - logw[k])): the resampler for
        log-densities, which do not underflow at large d the way the reference's densities do."""
        logw = _f64(logw).reshape(-1)
        N = logw.shape[0] if N is None else int(N)
        if seed is None:
            seed = _next_key()
        if N > logw.shape[0]:
            raise ValueError("N = %d exceeds the %d log-weights given" % (N, logw.shape[0]))
        a = np.empty(N, dtype=np.uint32)
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(v) for v in devices])
            check(_lib.lib().cusmc_metropolis_multi_host(devs, len(devices), _ptr(logw), N, int(B), int(seed), int(t), 1, _ptr(a)))
            return a
        ctx = ctx or default_context()
        check(_lib.lib().cusmc_metropolis_log_host(ctx._h, _ptr(logw), N, int(B), int(seed), int(t), _ptr(a)))
        return a

    @staticmethod
    def metropolis_hastings_log_dev(logw, a, B=10, t=1, seed=0, first=0, ctx=None):
        import torch
        ctx = ctx or default_context()
        _dev_vector(logw, "logw", (torch.float64,))
        _dev_vector(a, "a", _index_dtypes())
        check(_lib.lib().cusmc_metropolis_log_dev(ctx._h, C.c_void_p(logw.data_ptr()), logw.numel(), int(B),
                                                  int(seed), int(t), int(first), a.numel(),
                                                  C.c_void_p(a.data_ptr())))
        return a

    @staticmethod
    def metropolis_hastings_dev(w, a, B=10, t=1, seed=0, first=0, ctx=None):
        """Device-resident: w float64 CUDA tensor (all N weights), a int32/uint32-sized CUDA
        tensor receiving the ancestors of chains [first, first + len(a))."""
        import torch
        ctx = ctx or default_context()
        _dev_vector(w, "w", (torch.float64,))
        _dev_vector(a, "a", _index_dtypes())
        check(_lib.lib().cusmc_metropolis_dev(ctx._h, C.c_void_p(w.data_ptr()), w.numel(), int(B),
                                              int(seed), int(t), int(first), a.numel(),
                                              C.c_void_p(a.data_ptr())))
        return a


def propagate_dev(X_prev, a, G, Q, X_out, kind="mvn", nu=0.0, scale=1.0, seed=0, step=1, first=0, ctx=None):
    """propagate_K on device-resident tensors (src/mcmc.cpp:90-160):
    X_out[i] = [diag(c)] Q (scale xi) + G X_prev[a[i]] for the rows [first, first + len(X_out)).
    X_prev: N x d float64 CUDA tensor; a: int32 CUDA tensor of len(X_out) ancestors, or None."""
    ctx = ctx or default_context()
    d = _dev_matrix(X_prev, "X_prev").shape[1]
    _dev_matrix(X_out, "X_out", d)
    if a is not None:
        _dev_vector(a, "a", _index_dtypes(), X_out.shape[0])
    G, Q = _host_square(G, d, "G"), _host_square(Q, d, "Q")
    check(_lib.lib().cusmc_propagate_dev(ctx._h, _MVT if kind == "mvt" else _MVN, C.c_float(nu),
                                         C.c_void_p(X_prev.data_ptr()),
                                         None if a is None else C.c_void_p(a.data_ptr()), X_prev.shape[0], d,
                                         _ptr(G), _ptr(Q), float(scale), int(seed), int(step), int(first),
                                         X_out.shape[0], C.c_void_p(X_out.data_ptr())))
    return X_out


def pf_step_dev(obs, w_prev, X_prev, G, Q, y, F, a_out, X_out, w_out, kind="mvn", nu=0.0, B=10, scale=1.0,
                seed=0, step=1, first=0, log=False):
    """One time step of MCMC()'s loop (src/mcmc.cpp:292-308) on device-resident tensors:
    resample over w_prev -> propagate -> reweight against the observation distribution `obs`
    (pdf_{0,V}: its own mu is not used), for the rows [first, first + len(a_out)).  One launch for
    d <= 8, the three separate kernels otherwise; same numbers either way."""
    import torch
    d = obs.d
    _dev_matrix(X_prev, "X_prev", d)
    _dev_matrix(X_out, "X_out", d)
    _dev_vector(w_prev, "w_prev", (torch.float64,), X_prev.shape[0])
    _dev_vector(a_out, "a_out", _index_dtypes(), X_out.shape[0])
    _dev_vector(w_out, "w_out", (torch.float64,), X_out.shape[0])
    G, Q, y = _host_square(G, d, "G"), _host_square(Q, d, "Q"), _f64(y).reshape(-1)
    if y.shape[0] != d:
        raise ValueError("y must have %d entries" % d)
    Fm = None if F is None else _host_square(F, d, "F")
    check(_lib.lib().cusmc_pf_step_dev(obs._h, _MVT if kind == "mvt" else _MVN, C.c_float(nu),
                                       C.c_void_p(w_prev.data_ptr()), C.c_void_p(X_prev.data_ptr()),
                                       X_prev.shape[0], _ptr(G), _ptr(Q), _ptr(y), _ptr(Fm), int(B), float(scale),
                                       int(seed), int(step), int(first), a_out.numel(),
                                       C.c_void_p(a_out.data_ptr()), C.c_void_p(X_out.data_ptr()),
                                       C.c_void_p(w_out.data_ptr()), OUT_LOG if log else OUT_DENSITY))
    return a_out, X_out, w_out


def initialize_dev(m0, Q, X_out, kind="mvn", nu=0.0, scale=1.0, seed=0, first=0, ctx=None):
    """initialize() draws on a device-resident tensor (src/mcmc.cpp:44-88)."""
    ctx = ctx or default_context()
    m0 = _f64(m0).reshape(-1)
    d = m0.shape[0]
    _dev_matrix(X_out, "X_out", d)
    Q = _host_square(Q, d, "Q")
    check(_lib.lib().cusmc_initialize_dev(ctx._h, _MVT if kind == "mvt" else _MVN, C.c_float(nu), _ptr(m0),
                                          _ptr(Q), m0.shape[0], float(scale), int(seed), int(first),
                                          X_out.shape[0], C.c_void_p(X_out.data_ptr())))
    return X_out


# ---- R-level exports ----------------------------------------------------------------------------

def _batched_density(dist, x, d):
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        if x.shape[0] != d:
            raise ValueError("x has %d entries, mu has %d" % (x.shape[0], d))
        return float(dist.pdf_batch(x.reshape(1, d), None, log=False)[0])
    if x.ndim == 2 and x.shape[0] == d:  # d x N, columns = particles (R layout)
        return dist.pdf_batch(np.ascontiguousarray(x.T), None, log=False)
    raise ValueError("x must be a length-d vector or a d x N matrix")


# The R-level density calls build a distribution object per call (as the reference does:
# src/mvn_dist.rcpp.cpp:55).  A loop over particles with the same (mu, sigma) -- the reference's typical
# R usage -- would pay a factorisation, four device allocations and their release per particle, so the
# last few objects are kept: a repeated call is an upload of x, one launch and 8 bytes back.
_DIST_CACHE_SIZE = 4
_dist_cache = collections.OrderedDict()


def _cached_distribution(cls, mu, sigma, *nu):
    mu, sigma = _f64(mu), _f64(sigma)
    key = (cls.__name__, tuple(np.float32(v).tobytes() for v in nu), mu.tobytes(), sigma.shape, sigma.tobytes())
    dist = _dist_cache.get(key)
    if dist is not None:
        _dist_cache.move_to_end(key)
        return dist
    dist = cls(mu, sigma, *nu)
    _dist_cache[key] = dist
    while len(_dist_cache) > _DIST_CACHE_SIZE:
        _dist_cache.popitem(last=False)[1].close()
    return dist


def _clear_dist_cache():
    while _dist_cache:
        _dist_cache.popitem()[1].close()


# distribution handles point into their context: at interpreter exit they must go first, whatever
# order the module globals are torn down in
atexit.register(_clear_dist_cache)


def MVNPDF(x, mu, sigma):
    """double MVNPDF(x, mu, sigma) -- src/mvn_dist.rcpp.cpp:52-58: F = I; MVN(mu,sigma).pdf(x, F).
    Returns the DENSITY.  Batched extension: a d x N matrix x returns N densities."""
    dist = _cached_distribution(MultiVariateNormalDistribution, mu, sigma)
    return _batched_density(dist, x, dist.d)


def MVTPDF(x, mu, sigma, nu):
    """double MVTPDF(x, mu, sigma, nu) -- src/mvt_dist.rcpp.cpp:60-66."""
    dist = _cached_distribution(MultiVariateTStudentDistribution, mu, sigma, nu)
    return _batched_density(dist, x, dist.d)


_eigen_cache = collections.OrderedDict()  # sigma -> Q, for repeated R-level draws from one covariance


def _cached_eigen_sqrt(sigma):
    key = (sigma.shape, sigma.tobytes())
    Q = _eigen_cache.get(key)
    if Q is None:
        Q = _eigen_cache[key] = eigenSolver(sigma)
        while len(_eigen_cache) > _DIST_CACHE_SIZE:
            _eigen_cache.popitem(last=False)
    else:
        _eigen_cache.move_to_end(key)
    return Q


def MVN(mu, sigma, compat=False):
    """VectorXd MVN(mu, sigma) -- src/mvn_dist.rcpp.cpp:31-37.  The reference passes `sigma`
    ITSELF as the square-root factor Q (:35) and inflates the variance 3x (F6); compat=True
    reproduces that distribution, the default draws from N(mu, sigma)."""
    dist = _cached_distribution(MultiVariateNormalDistribution, mu, sigma)
    Q = dist.sigma if compat else _cached_eigen_sqrt(dist.sigma)
    return dist.sample(Q, 200, compat=compat)[0]


def MVT(mu, sigma, nu, compat=False):
    """VectorXd MVT(mu, sigma, nu) -- src/mvt_dist.rcpp.cpp:28-49 (Q = eigen square root)."""
    dist = _cached_distribution(MultiVariateTStudentDistribution, mu, sigma, nu)
    return dist.sample(_cached_eigen_sqrt(dist.sigma), 200, compat=compat)[0]


def metropolis_hastings(w, N, B):
    """VectorXd metropolis_hastings(w, N, B) -- src/samplers.rcpp.cpp:35-55: t = 1, returns the
    N ancestors as DOUBLES, 0-based, like the reference."""
    # t = 1 as the reference's export passes it (src/samplers.rcpp.cpp:43); a fresh key per call
    return Sampler.metropolis_hastings(w, N, t=1, B=B, seed=_next_key()).astype(np.float64)


def run(N, d, timeSteps, Y, m0, C0, F, G, V, W, df, resampler, distribution, p=0, B=10,
        compat=False, seed=None, write_csv=False, return_ancestors=False, devices=None):
    """List run(N, d, timeSteps, Y, m0, C0, F, G, V, W, df, resampler, distribution, p)
    -- src/run.rcpp.cpp:58-126.  Returns {"weights": T x N, "posterior_x": T x N x d}.
    df reaches the filter as df (the reference passes it in the wrong slot: SURVEY.md F8).
    B = 10 is the reference's hard-coded value (src/mcmc.cpp:291).
    devices = [0, 1, ...] shards the particles over those GPUs below the C ABI
    (cusmc_pf_run_multi_host; same numbers as one GPU); the environment variable CUSMC_DEVICES="0,1,..."
    does the same for callers that cannot pass the argument (the R package)."""
    N, d, T = int(N), int(d), int(timeSteps)
    if not 0 <= int(p) < N:
        raise ValueError("p = %d must satisfy 0 <= p < N" % p)  # assert(p < N): run.rcpp.cpp:64
    Y = _f64(Y)
    if Y.shape != (d, T):
        raise ValueError("Y must be d x timeSteps (columns = time), got %s" % (Y.shape,))
    Yt = np.ascontiguousarray(Y.T)
    m0, C0, F, G, V, W = (_f64(a) for a in (m0, C0, F, G, V, W))
    if seed is None:
        seed = _next_key()
    for name, m in (("C0", C0), ("F", F), ("G", G), ("V", V), ("W", W)):
        if m.shape != (d, d):
            raise ValueError("%s must be %d x %d, got %s" % (name, d, d, m.shape))
    if m0.reshape(-1).shape[0] != d:
        raise ValueError("m0 must have %d entries" % d)
    X = np.empty((T, N, d))
    w = np.empty((T, N))
    a = np.empty((T, N), dtype=np.uint32) if return_ancestors else None  # (not copied back unless asked for)
    tail = (_ptr(Yt), N, d, T, _ptr(m0), _ptr(C0), _ptr(F), _ptr(G), _ptr(V), _ptr(W), C.c_float(df),
            str(resampler).encode(), str(distribution).encode(), int(B), SQRT3 if compat else 1.0, int(seed),
            _ptr(X), _ptr(w), _ptr(a))
    if devices is not None:
        devs = (C.c_int * len(devices))(*[int(v) for v in devices])
        check(_lib.lib().cusmc_pf_run_multi_host(devs, len(devices), *tail))
    else:
        check(_lib.lib().cusmc_pf_run_host(default_context()._h, *tail))
    if write_csv:
        from .io import writeOutput
        writeOutput(Yt, w, X, N, d, T, int(p))
    out = {"weights": w, "posterior_x": X}
    if return_ancestors:
        out["ancestors"] = a
    return out
