"""CPU suite, part 1: the oracle (oracle/cusmc_oracle.c) against the reference's own known
answers, the Random123 known-answer vectors, the committed golden fixtures and scipy.
No GPU, no libcusmc_hip compute."""
import numpy as np
import pytest

from conftest import DIMS, RESAMPLE_CASES, spd


# --- the three values the reference publishes -----------------------------------------------

def test_known_answer_mvnpdf(oracle):
    # CuSMC/CuSMC.tex:95-105, man/MVNPDF.Rd:22-27: MVNPDF(c(0,0), c(0,0), diag(2)) = 0.1591549
    v = oracle.mvn_pdf([0, 0], [0, 0], np.eye(2), np.eye(2))
    assert abs(v - 0.1591549) < 5e-8
    assert abs(v - 1 / (2 * np.pi)) < 1e-16


def test_known_answer_mvtpdf(oracle):
    # CuSMC/CuSMC.tex:131-142: MVTPDF(c(0,0,0), c(0,0,0), diag(3), 3.0) = 0.07799708
    v = oracle.mvt_pdf([0, 0, 0], [0, 0, 0], np.eye(3), 3.0, np.eye(3))
    assert abs(v - 0.07799708) < 5e-9
    assert abs(v - 0.0779970835340203) < 1e-15


def test_known_answer_metropolis_zero_weights(oracle):
    # man/metropolis_hastings.Rd:22-27: w = c(0,0), N = 2, B = 10 -> c(0, 1) (0/0 never accepts)
    for seed in (0, 1, 12345):
        assert oracle.metropolis(np.zeros(2), 10, seed).tolist() == [0, 1]


# --- RNG contract ------------------------------------------------------------------------------

@pytest.mark.parametrize("ctr,key,expect", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_known_answers(oracle, ctr, key, expect):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox4x32_10(ctr, key).tolist() == expect


# --- dense LU (what Eigen's determinant()/inverse() do) ------------------------------------------

@pytest.mark.parametrize("d", [1, 2, 5, 17, 64])
def test_lu_det_inverse(oracle, d):
    rng = np.random.default_rng(d)
    S = rng.standard_normal((d, d)) + d * np.eye(d)
    assert np.isclose(oracle.det(S), np.linalg.det(S), rtol=1e-10)
    assert np.allclose(oracle.inverse(S) @ S, np.eye(d), atol=1e-10)


# --- densities against the fixtures (scipy) ------------------------------------------------------

@pytest.mark.parametrize("d", DIMS)
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_pdf_matches_golden(oracle, golden, d, dist):
    g = lambda k: golden["pdf_d%d_%s" % (d, k)]
    nu = float(g("nu"))
    p = oracle.pdf_batch(g("X"), g("mu"), g("sigma"), g("F"), dist, nu)
    assert np.array_equal(p, g(dist + "_pdf_oracle"))  # the restatement itself has not drifted
    assert np.allclose(np.log(p), g(dist + "_logpdf_scipy"), rtol=1e-9, atol=1e-9)
    w = oracle.reweight(g("X"), g("y"), g("F"), g("sigma"), dist, nu)
    assert np.array_equal(w, g(dist + "_reweight_oracle"))
    assert np.allclose(np.log(w), g(dist + "_reweight_scipy"), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("d", DIMS)
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_hoisted_form_equals_faithful_form(oracle, golden, d, dist):
    g = lambda k: golden["pdf_d%d_%s" % (d, k)]
    nu = float(g("nu"))
    lp = oracle.logpdf_hoisted(g("X"), g("mu"), g("sigma"), g("F"), dist, nu)
    assert np.allclose(lp, g(dist + "_logpdf_scipy"), rtol=1e-10, atol=1e-10)


def test_norms(oracle):
    rng = np.random.default_rng(3)
    S = spd(rng, 6)
    assert np.isclose(oracle.mvn_norm(S), (2 * np.pi) ** -3 * np.linalg.det(S) ** -0.5, rtol=1e-12)
    from scipy.special import gammaln
    nu = 2.5
    ln = gammaln((nu + 6) / 2) - gammaln(nu / 2) - 3 * np.log(np.pi * nu) - 0.5 * np.log(np.linalg.det(S))
    assert np.isclose(np.log(oracle.mvt_norm(S, nu)), ln, rtol=1e-12)


def test_pdf_one_arg_overload_ignores_mu(oracle):
    # pdf(y) uses y itself in the quadratic form (src/statistics.cc.cpp:171-180)
    S = spd(np.random.default_rng(5), 4)
    y = np.array([0.3, -1.0, 0.2, 0.9])
    assert oracle.mvn_pdf(y, np.ones(4), S, None) == oracle.mvn_pdf(y, np.zeros(4), S, np.eye(4))


def test_not_spd_is_reported(oracle):
    with pytest.raises(ValueError):
        oracle.logpdf_hoisted(np.zeros((1, 2)), None, np.array([[1.0, 2.0], [2.0, 1.0]]))


# --- resampler -----------------------------------------------------------------------------------

@pytest.mark.parametrize("name", RESAMPLE_CASES)
@pytest.mark.parametrize("B", [1, 10, 37])
def test_resampler_matches_golden(oracle, golden, name, B):
    w = golden["resample_%s_w" % name]
    a = oracle.metropolis(w, B, seed=20240 + B, step=1)
    assert a.dtype == np.uint32 and a.shape == w.shape
    assert np.array_equal(a, golden["resample_%s_B%d" % (name, B)])
    assert a.max() < w.shape[0]


def _python_chain(oracle, w, N, B, seed, step, chains=None):
    """Pure-Python restatement of src/samplers.cpp:21-35 under RNG contract 3, in exact integer / rational arithmetic
    (Fraction), sharing nothing with the C oracle but the Philox blocks: one block per two steps; u = the uniform real
    whose leading 32 bits are word 2h, completed by 53 more bits (domain 7) only when the ratio lies inside its
    cell; j by Lemire's unbiased multiply-and-reject on word 2h + 1 (redraws: domains 16, 17, ...)."""
    from fractions import Fraction
    key = [seed & 0xFFFFFFFF, seed >> 32]
    tN = (2 ** 32 - N) % N
    expect, refined, redrawn = [], 0, 0
    for i in (range(N) if chains is None else chains):
        k = i
        for n in range(B):
            r = oracle.philox4x32_10([i, n >> 1, step, 1], key)
            a, b = int(r[2 * (n & 1)]), int(r[2 * (n & 1) + 1])
            m = b * N
            if (m & 0xFFFFFFFF) < tN:
                redrawn += 1
                q, found = 0, False
                while not found:
                    x = oracle.philox4x32_10([i, n, step, 16 + q], key)
                    for c in range(4):
                        m = int(x[c]) * N
                        if (m & 0xFFFFFFFF) >= tN:
                            found = True
                            break
                    q += 1
            j = m >> 32
            ratio = w[j] / w[k]                       # the reference's double division
            if ratio != ratio:
                continue
            rr = Fraction(ratio) if np.isfinite(ratio) else (Fraction(10) ** 400 if ratio > 0 else -Fraction(10) ** 400)
            lo, hi = Fraction(a, 2 ** 32), Fraction(a + 1, 2 ** 32)
            if hi <= rr:
                acc = True
            elif lo > rr:
                acc = False
            else:
                refined += 1
                x = oracle.philox4x32_10([i, n, step, 7], key)
                v = Fraction((int(x[0]) << 32 | int(x[1])) >> 11, 2 ** 53)
                acc = lo + v / 2 ** 32 <= rr          # u = (a + V) 2^-32 <= ratio
            if acc:
                k = j
        expect.append(k)
    return expect, refined, redrawn


def test_resampler_semantics(oracle):
    rng = np.random.default_rng(11)
    w = rng.random(50)
    N, B, seed, step = 50, 7, 0xDEADBEEFCAFE, 3
    expect, _, _ = _python_chain(oracle, w, N, B, seed, step)
    assert oracle.metropolis(w, B, seed, step).tolist() == expect


def test_resampler_refinement_forced(oracle):
    """Contract 3's first completion, which a random run meets once in 2^32 steps: the ratio INSIDE u's 32-bit cell
    (w[1] / w[0] = (a + 1/2) 2^-32 for the very a chain 0's first step draws), so that the 53 refinement bits decide."""
    hits = 0
    for seed in range(40):
        step = 2
        r = oracle.philox4x32_10([0, 0, step, 1], [seed, 0])
        a, j = int(r[0]), (int(r[1]) * 2) >> 32
        if j != 1:
            continue  # (a self-proposal: ratio 1, accepted outright)
        w = np.array([1.0, (a + 0.5) * 2.0 ** -32])
        expect, refined, _ = _python_chain(oracle, w, 2, 1, seed, step)
        assert refined >= 1
        assert oracle.metropolis(w, 1, seed, step).tolist() == expect
        hits += 1
    assert hits >= 5


def test_resampler_index_redraws(oracle):
    """Contract 3's second completion: N = 3 * 2^20 leaves tN = 2^20, i.e. one index candidate in 4096 is redrawn
    (Lemire's rejection); the exact-arithmetic restatement over the first 60000 chains meets a few dozen."""
    N, B, seed, step = 3 * 2 ** 20, 3, 31337, 4
    w = np.random.default_rng(8).random(N)
    sub = 60000
    expect, _, redrawn = _python_chain(oracle, w, N, B, seed, step, chains=range(sub))
    assert redrawn >= 10
    assert oracle.metropolis(w, B, seed, step)[:sub].tolist() == expect


def test_resampler_index_is_uniform(oracle):
    """B = 1 over equal weights always accepts: the ancestors ARE the index draws."""
    from scipy import stats
    N = 1000
    counts = np.zeros(N)
    for seed in range(20):
        counts += np.bincount(oracle.metropolis(np.ones(N), 1, seed=seed), minlength=N)
    assert stats.chisquare(counts).pvalue > 1e-3


def test_resampler_targets_weights(oracle):
    # long chains forget the start: ancestor frequencies follow w (Murray-Lee-Jacob)
    w = np.array([1.0, 2.0, 3.0, 4.0] * 250)
    a = oracle.metropolis(w, 200, seed=5)
    freq = np.bincount(w[a].astype(int), minlength=5)[1:] / a.size
    assert np.allclose(freq, [0.1, 0.2, 0.3, 0.4], atol=0.05)


def test_resampler_zero_and_dominant(oracle):
    assert np.array_equal(oracle.metropolis(np.zeros(33), 10, 1), np.arange(33))
    w = np.r_[np.full(99, 1e-300), 1.0]
    assert np.mean(oracle.metropolis(w, 400, 2) == 99) > 0.95


# --- draws ---------------------------------------------------------------------------------------

def test_eigen_sqrt(oracle):
    S = spd(np.random.default_rng(8), 9)
    Q = oracle.eigen_sqrt(S)
    assert np.allclose(Q @ Q.T, S, atol=1e-12)


@pytest.mark.parametrize("scale,var", [(1.0, 1.0), (3 ** 0.5, 3.0)])
def test_mvn_draw_moments(oracle, scale, var):
    # scale sqrt(3) = the distribution of the reference's CPU transform
    S = np.array([[2.0, 0.6], [0.6, 1.0]])
    X, w = oracle.initialize(200000, [1.0, -2.0], oracle.eigen_sqrt(S), scale=scale, seed=4)
    assert np.allclose(X.mean(0), [1.0, -2.0], atol=0.02)
    assert np.allclose(np.cov(X.T), var * S, atol=0.03 * var)
    assert np.allclose(w, 1 / 200000)


def test_mvt_draw_componentwise_chi(oracle):
    # each component gets its OWN sqrt(nu/chi2): marginals are t_nu, var nu/(nu-2)
    X, _ = oracle.initialize(200000, [0.0, 0.0], np.eye(2), "mvt", 5.0, seed=9)
    assert np.allclose(np.var(X, axis=0), 5 / 3, atol=0.05)
    assert abs(np.corrcoef(X.T)[0, 1]) < 0.01


def test_propagate_is_gather_plus_draw(oracle):
    rng = np.random.default_rng(2)
    Xp = rng.standard_normal((32, 3))
    a = rng.integers(0, 32, 32).astype(np.uint32)
    G = rng.standard_normal((3, 3))
    X = oracle.propagate(Xp, a, G, np.zeros((3, 3)), seed=1, step=4)  # Q = 0: pure G x[a]
    assert np.allclose(X, Xp[a] @ G.T, atol=1e-14)


def test_filter_matches_golden(oracle, golden):
    I = np.eye(2)
    for dist, nu in (("mvn", 0.0), ("mvt", 5.0)):
        X, w, a = oracle.pf_run(golden["pf_y"], 64, np.zeros(2), I, I, I, 0.5 * I, 0.1 * I, dist, nu,
                                B=10, seed=99)
        assert np.array_equal(a, golden["pf_%s_a" % dist])
        assert np.allclose(X, golden["pf_%s_X" % dist], rtol=0, atol=1e-12)
        assert np.allclose(w, golden["pf_%s_w" % dist], rtol=1e-10)
        assert np.all(a[0] == 0) and np.allclose(w[0], 1 / 64)


def test_log_weight_resampler_restatement(oracle):
    """oracle_metropolis_log is the density chain with w = exp(lw): identical ancestors wherever
    no accept test sits within rounding of its boundary (all but a handful of 1e5 chains), and it keeps
    working where the densities underflow to 0/0.  exp_nonpos agrees with libm to 2 ulp."""
    import math
    rng = np.random.default_rng(11)
    for t in np.concatenate([-np.logspace(-12, 2.8, 400), [0.0, -1e-300]]):
        assert abs(oracle.exp_nonpos(t) - math.exp(t)) <= 4e-16 * math.exp(t)
    assert oracle.exp_nonpos(-800.0) == 0.0 and oracle.exp_nonpos(-np.inf) == 0.0 and math.isnan(oracle.exp_nonpos(float("nan")))
    N, B = 100_000, 10
    lw = -0.5 * rng.chisquare(32, N) - 40.0
    a_log = oracle.metropolis_log(lw, B, 7, step=3)
    a_den = oracle.metropolis(np.exp(lw), B, 7, step=3)
    assert np.mean(a_log != a_den) < 1e-4
    deep = lw - 2000.0                      # exp() of these is 0: the density chain would see 0/0 everywhere
    assert np.array_equal(oracle.metropolis_log(deep, B, 7, step=3), a_log)   # only differences matter
    assert np.array_equal(oracle.metropolis(np.exp(deep), B, 7, step=3), np.arange(N))  # NaN ratios never accept
    lw2 = lw.copy(); lw2[::5] = -np.inf    # zero weights: never entered
    a2 = oracle.metropolis_log(lw2, B, 7, step=3)
    moved = a2 != np.arange(N)
    assert not np.isinf(lw2[a2[moved]]).any()


def _random_covariances(rng, N, d):
    A = rng.standard_normal((N, d, d))
    scale = np.exp(rng.uniform(-2, 2, size=(N, 1, 1)))
    return scale * (A @ np.transpose(A, (0, 2, 1)) / d + np.eye(d))


@pytest.mark.parametrize("d", [1, 2, 3, 8, 16])
def test_per_particle_covariances_restatement(oracle, d):
    """SURVEY.md 8(f) row 4.  The per-covariance density is the reference's pdf() with a
    distribution object per particle (LU determinant + inverse per call, src/statistics.cc.cpp:
    171-196, 295-324): checked against scipy; the batched Cholesky (the kernels' operation order)
    against numpy."""
    from scipy import stats
    rng = np.random.default_rng(100 + d)
    N = 40
    S = _random_covariances(rng, N, d)
    X = rng.standard_normal((N, d))
    mu = rng.standard_normal((N, d))
    L, logdet, info = oracle.chol_batched(S)
    assert not info.any()
    for i in range(N):
        assert np.allclose(L[i], np.linalg.cholesky(S[i]), rtol=1e-13, atol=1e-14 * np.abs(L[i]).max())
        assert np.isclose(logdet[i], np.linalg.slogdet(S[i])[1], rtol=1e-12, atol=1e-12)
    p = oracle.pdf_percov(X, mu, S)
    want = np.array([stats.multivariate_normal(mu[i], S[i]).pdf(X[i]) for i in range(N)])
    assert np.allclose(p, want, rtol=1e-10)
    pt = oracle.pdf_percov(X, mu[0], S, dist="mvt", nu=4.0)
    want = np.array([stats.multivariate_t(mu[0], S[i], df=4.0).pdf(X[i]) for i in range(N)])
    assert np.allclose(pt, want, rtol=1e-10)
    # a covariance that is not positive definite is reported, with the index of the pivot
    S[3] = np.eye(d)
    S[3][d - 1, d - 1] = -1.0
    assert oracle.chol_batched(S)[2][3] == d
