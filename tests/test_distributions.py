"""Distributional parity of the RNG-driven functions with the reference's semantics.

The reference's draws cannot be seeded (std::random_device per call: src/samplers.cpp:10-11,
src/statistics.cc.cpp:231-232), so for sample(), metropolis_hastings(), initialize(), propagate_K() and the
filter built from them the only parity that exists WITH THE REFERENCE is distributional:

  * MVN::sample (src/statistics.cc.cpp:224-259): mu + Q xi, xi ~ N(0, I) on the device path
    (src/mvn_dist.cu.cpp:24-31), N(0, 3 I) from the CPU path's 200-term sum ("compat");
  * MVT::sample (src/statistics.cc.cpp:355-412): every COMPONENT of Q xi scaled by its own
    sqrt(nu / chi2_nu) -- with Q = I the marginals are independent Student-t_nu;
  * Sampler::metropolis_hastings (src/samplers.cpp:21-35): an independence Metropolis chain on the particle
    indices with uniform proposals, stationary law w / sum(w);
  * particle_filter() on a linear-Gaussian model: the weighted particle mean estimates E[x_t | y_1..t], which
    the Kalman recursion gives exactly.

These tests hold the oracle (CPU) and the HIP path (-m gpu, through the C ABI) against those EXACT laws with
scipy -- a check that shares no arithmetic with the oracle.  Seeds are fixed, so the outcomes are deterministic;
the thresholds are p > 1e-3 for goodness-of-fit tests (a correct sampler fails one such test in a thousand
seeds; each check is paired with a slightly wrong law that it must reject, so it has power at these sample sizes).
"""
import numpy as np
import pytest
from scipy import stats

P_MIN = 1e-3


def ks_p(x, dist):
    return stats.kstest(x, dist.cdf).pvalue


def check_normals(draw):
    """draw(count, d, compat, seed) -> count x d draws with mu = 0, Q = I"""
    for compat, var in ((False, 1.0), (True, 3.0)):
        x = draw(125_000, 8, compat, 11).reshape(-1)
        assert ks_p(x, stats.norm(0.0, var ** 0.5)) > P_MIN, ("N(0,%g)" % var)
        assert ks_p(x, stats.norm(0.0, (1.03 * var) ** 0.5)) < 1e-6       # (the test can tell 3 % in the variance)
        # pairs come from one Box-Muller evaluation: no correlation between the two outputs, nor between neighbours
        X = x.reshape(-1, 8)
        c = np.corrcoef(X.T)
        assert np.max(np.abs(c - np.eye(8))) < 0.012


def check_covariance(draw_q, eigen_sqrt):
    """draw_q(count, mu, Q, seed) -> draws of N(mu, Q Q^T)"""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((4, 4))
    S = A @ A.T / 4 + np.eye(4)
    mu = np.array([1.0, -2.0, 0.5, 3.0])
    X = draw_q(400_000, mu, eigen_sqrt(S), 12)
    n = X.shape[0]
    assert np.all(np.abs(X.mean(0) - mu) < 5 * np.sqrt(np.diag(S) / n))
    # Var(s_ij) = (S_ii S_jj + S_ij^2) / n for a Gaussian sample
    tol = 5 * np.sqrt((np.outer(np.diag(S), np.diag(S)) + S * S) / n)
    assert np.all(np.abs(np.cov(X.T) - S) < tol)


def check_student(draw_t):
    """draw_t(count, nu, seed) -> count x 2 draws with mu = 0, Q = I: two INDEPENDENT t_nu marginals (F7)"""
    # 0.75, 1.5: the a < 1 boost of the gamma sampler (src/mvt_dist.cu.cpp:53-60); 2.5, 30: Marsaglia-Tsang; the
    # integers: RNG contract 4's closed forms (1: a squared normal alone; 3, 7: uniforms and a squared normal)
    for nu in (0.75, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 7.0, 30.0):
        X = draw_t(200_000, nu, 13)
        for j in range(2):
            assert ks_p(X[:, j], stats.t(nu)) > P_MIN, (nu, j)
        assert ks_p(X[:, 0], stats.t(1.25 * nu)) < 1e-3 or nu >= 30.0      # (25 % in nu is visible below nu = 30)
        # a chi-square per component: |x_0| and |x_1| are independent (one shared chi-square would couple them:
        # Spearman rho ~ 0.27 at nu = 3)
        rho = stats.spearmanr(np.abs(X[:, 0]), np.abs(X[:, 1])).statistic
        assert abs(rho) < 0.01, (nu, rho)


def check_resampler(resample):
    """resample(w, B, seed) -> ancestors of len(w) chains.  After B = 300 steps of the independence chain (geometric
    rate 1 - mean(w)/max(w) ~ 0.94 here: 1e-8 left) the ancestors are draws from w / sum(w): chi-square
    goodness of fit over 40 bins of equal probability mass, atoms ordered by weight."""
    rng = np.random.default_rng(21)
    N = 100_000
    w = np.exp(-0.5 * rng.chisquare(8, N)) * 1e-12     # (unnormalised, tiny: only ratios matter)
    a = resample(w, 300, 77)
    order = np.argsort(w)
    cum = np.cumsum(w[order]) / w.sum()
    bin_of_rank = np.minimum((cum * 40).astype(int), 39)
    bin_of_atom = np.empty(N, dtype=int)
    bin_of_atom[order] = bin_of_rank
    expected = np.bincount(bin_of_atom, weights=w, minlength=40) / w.sum() * N
    observed = np.bincount(bin_of_atom[a], minlength=40)
    assert stats.chisquare(observed, expected).pvalue > P_MIN
    # and B = 2 is far from it (the chain starts at i, uniform): the test has power
    a2 = resample(w, 2, 77)
    assert stats.chisquare(np.bincount(bin_of_atom[a2], minlength=40), expected).pvalue < 1e-6
    # zero weights are never selected once the chain has left them
    w0 = w.copy()
    w0[::3] = 0.0
    a0 = resample(w0, 300, 78)
    assert np.all(w0[a0] > 0.0)


def kalman_means(Y, m0, C0, F, G, V, W):
    """E[x_t | y_1..t] for t = 1..T-1 (the filter does not use y_0: w_0 = 1/N, src/mcmc.cpp:85)."""
    m, P = m0.copy(), C0.copy()
    out = []
    for t in range(1, Y.shape[0]):
        m, P = G @ m, G @ P @ G.T + W
        S = F @ P @ F.T + V
        K = P @ F.T @ np.linalg.inv(S)
        m = m + K @ (Y[t] - F @ m)
        P = P - K @ F @ P
        out.append(m.copy())
    return np.array(out)


def lg_model(T=12):
    rng = np.random.default_rng(31)
    G = np.array([[0.9, 0.2], [-0.1, 0.8]])
    F = np.array([[1.0, 0.5], [0.0, 1.0]])
    W = np.array([[0.3, 0.05], [0.05, 0.2]])
    V = np.array([[0.5, 0.1], [0.1, 0.4]])
    m0, C0 = np.array([1.0, -1.0]), np.eye(2)
    x = m0 + rng.standard_normal(2)
    Y = np.zeros((T, 2))
    for t in range(T):
        if t:
            x = G @ x + np.linalg.cholesky(W) @ rng.standard_normal(2)
        Y[t] = F @ x + np.linalg.cholesky(V) @ rng.standard_normal(2)
    return Y, m0, C0, F, G, V, W


def check_filter(X, w, model, tol):
    est = (w[1:, :, None] * X[1:]).sum(1) / w[1:].sum(1)[:, None]
    assert np.max(np.abs(est - kalman_means(*model))) < tol


# ---- the oracle (CPU) ---------------------------------------------------------------------------------------------

def test_oracle_normals(oracle):
    check_normals(lambda n, d, compat, seed: oracle.initialize(n, np.zeros(d), np.eye(d), "mvn", 0.0,
                                                               3 ** 0.5 if compat else 1.0, seed=seed)[0])


def test_oracle_covariance(oracle):
    check_covariance(lambda n, mu, Q, seed: oracle.initialize(n, mu, Q, seed=seed)[0], oracle.eigen_sqrt)


def test_oracle_student_marginals(oracle):
    check_student(lambda n, nu, seed: oracle.initialize(n, np.zeros(2), np.eye(2), "mvt", nu, seed=seed)[0])


def check_chi_square(draw):
    """draw(count, d, nu, seed) -> count x d chi-square draws: every component chi^2_nu, components independent --
    in particular the two halves of a component pair, which the RNG contract feeds from the same Philox blocks."""
    # (integer nu <= 16: closed forms -- products of 1 .. 8 uniforms, a squared normal on top for odd nu; 17 and
    # the fractional ones: Marsaglia-Tsang)
    for nu in (0.5, 1.0, 2.0, 2.5, 3.0, 4.0, 5.0, 6.0, 7.0, 11.0, 15.0, 16.0, 17.0, 30.0):
        x = draw(150_000, 5, nu, 23)
        for j in range(5):
            assert ks_p(x[:, j], stats.chi2(nu)) > P_MIN, (nu, j)
        assert ks_p(x[:, 0], stats.chi2(1.04 * nu)) < 1e-4, nu       # (4 % in nu is visible)
        for i, j in ((0, 1), (2, 3), (1, 2), (3, 4)):                 # pair mates and neighbours across pairs
            rho = stats.spearmanr(x[:, i], x[:, j]).statistic
            assert abs(rho) < 0.012, (nu, i, j, rho)


def test_oracle_chi_square_law(oracle):
    assert oracle.rng_contract() == 4
    check_chi_square(lambda n, d, nu, seed: oracle.chi_square(n, d, nu, seed=seed, step=3))


def test_oracle_resampler_stationary_law(oracle):
    check_resampler(lambda w, B, seed: oracle.metropolis(w, B, seed))


def test_oracle_filter_against_kalman(oracle):
    model = lg_model()
    X, w, _ = oracle.pf_run(model[0], 100_000, *model[1:], B=40, seed=5)
    check_filter(X, w, model, 0.03)


# ---- the HIP path, through the C ABI (-m gpu) ---------------------------------------------------------------------

@pytest.fixture(scope="module")
def cs():
    import cusmc_amd
    from cusmc_amd import _lib
    assert _lib.lib().cusmc_device_count() > 0, "no GPU visible: the gpu suite needs an MI355X"
    return cusmc_amd


@pytest.mark.gpu
def test_gpu_normals(cs):
    def draw(n, d, compat, seed):
        D = cs.MultiVariateNormalDistribution(np.zeros(d), np.eye(d))
        try:
            return D.sample(np.eye(d), count=n, compat=compat, seed=seed, step=1)
        finally:
            D.close()
    check_normals(draw)


@pytest.mark.gpu
def test_gpu_covariance(cs):
    def draw(n, mu, Q, seed):
        D = cs.MultiVariateNormalDistribution(mu, Q @ Q.T)
        try:
            return D.sample(Q, count=n, seed=seed, step=2)
        finally:
            D.close()
    check_covariance(draw, np.linalg.cholesky)  # (any Q with Q Q^T = S)


@pytest.mark.gpu
def test_gpu_student_marginals(cs):
    def draw(n, nu, seed):
        D = cs.MultiVariateTStudentDistribution(np.zeros(2), np.eye(2), nu)
        try:
            return D.sample(np.eye(2), count=n, seed=seed, step=3)
        finally:
            D.close()
    check_student(draw)


@pytest.mark.gpu
@pytest.mark.parametrize("nu", [4.0, 2.0, 3.0, 2.5])
@pytest.mark.parametrize("d", [16, 64, 256])
def test_gpu_student_marginals_matrix_core_kernels(cs, d, nu):
    """The same law out of the matrix-core proposal kernels (d = 16, 64: propagate_mfma_kernel; d = 256:
    propagate_wide_kernel): two of the d components tested per kernel, chi-square per component -- nu = 4, 2: the
    closed forms of even nu, 3: a closed form of odd nu, 2.5: Marsaglia-Tsang, all through the lane exchange of the C layout (chi_square_clayout).
    Q is block diagonal with a rotation in every 2 x 2 block (Q Q^T = I, but NOT diagonal: a diagonal Q would
    take propagate_diag_kernel)."""
    c, s = np.cos(0.3), np.sin(0.3)
    Q = np.kron(np.eye(d // 2), np.array([[c, -s], [s, c]]))
    D = cs.MultiVariateTStudentDistribution(np.zeros(d), np.eye(d), nu)
    X = D.sample(Q, count=100_000, seed=17, step=4)
    D.close()
    for j in (0, 1, d - 1):
        assert ks_p(X[:, j], stats.t(nu)) > P_MIN, j
    assert abs(stats.spearmanr(np.abs(X[:, 0]), np.abs(X[:, 1])).statistic) < 0.012
    assert abs(stats.spearmanr(np.abs(X[:, 0]), np.abs(X[:, 4])).statistic) < 0.012
    assert ks_p(X[:, d // 2], stats.norm()) < 1e-6


@pytest.mark.gpu
def test_gpu_chi_square_law_through_the_draws(cs):
    """chi^2 itself, recovered from Student-t draws with Q = I, mu = 0 and the SAME seed's Normal draws: the
    contract keys the normals identically for both kinds, so x_t / x_n = sqrt(nu / chi2) component by component."""
    for nu in (2.0, 4.0, 2.5, 0.5):
        T = cs.MultiVariateTStudentDistribution(np.zeros(6), np.eye(6), nu)
        Nn = cs.MultiVariateNormalDistribution(np.zeros(6), np.eye(6))
        xt = T.sample(np.eye(6), count=150_000, seed=29, step=2)
        xn = Nn.sample(np.eye(6), count=150_000, seed=29, step=2)
        T.close(); Nn.close()
        chi = nu * (xn / xt) ** 2
        for j in range(6):
            assert ks_p(chi[:, j], stats.chi2(nu)) > P_MIN, (nu, j)
        assert abs(stats.spearmanr(chi[:, 0], chi[:, 1]).statistic) < 0.012


@pytest.mark.gpu
def test_gpu_resampler_stationary_law(cs):
    check_resampler(lambda w, B, seed: cs.Sampler.metropolis_hastings(w, B=B, seed=seed))


@pytest.mark.gpu
def test_gpu_filter_against_kalman(cs):
    Y, m0, C0, F, G, V, W = model = lg_model()
    out = cs.run(400_000, 2, Y.shape[0], Y.T, m0, C0, F, G, V, W, 0.0, "metropolis", "mvn", B=40, seed=5)
    check_filter(out["posterior_x"], out["weights"], model, 0.02)
