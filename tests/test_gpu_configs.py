"""GPU suite, part 2: every BASELINE.json config AT ITS SIZE ("configs not exercised at their
size"), through the C ABI, against the oracle on the GPU box's host cores and through size-independent
properties.  configs[0] (1e4 x d = 8, the CPU plumbing case) and the headline (1e6 x 64) live in
test_gpu_parity.py; here:

  configs[1]  metropolis_hastings(): 1e5 chains, weights = d = 32 MVN densities, 1e3 iterations
  configs[2]  particle filter: 1e6 particles, T = 100, linear-Gaussian, MVN likelihood
  configs[3]  MVT (nu = 4) target, 1e6 chains, d = 64, 1e4 iterations: one GPU's share of 8 (1.25e5 chains)
  configs[4]  d = 256 MVN, 4e6 particles: one GPU's share of 8 (5e5 particles)
"""
import numpy as np
import pytest

from conftest import spd

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def cs():
    import cusmc_amd
    from cusmc_amd import _lib
    assert _lib.lib().cusmc_device_count() > 0, "no GPU visible: the gpu suite needs an MI355X"
    return cusmc_amd


def rel_err(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


def test_config1_metropolis_1e5_chains_1e3_iterations_bit_exact(cs, oracle):
    """configs[1] at its size: 1e8 accept/reject steps, every ancestor equal to the oracle's chain
    (src/samplers.cpp:21-35 under the Philox contract).  The weights are d = 32 MVN densities (~1e-20:
    only the ratio form w[j]/w[k] keeps them apart), computed on the GPU and fed to both sides."""
    N, B, d = 100_000, 1000, 32
    rng = np.random.default_rng(2)
    sigma = spd(rng, d)
    X = rng.standard_normal((N, d)) @ np.linalg.cholesky(sigma).T
    D = cs.MultiVariateNormalDistribution(None, sigma)
    w = D.pdf_batch(X, log=False)
    assert rel_err(np.log(w[:512]), oracle.logpdf_hoisted(X[:512], None, sigma)) < RTOL
    seed = 0xC0FFEE1234
    a = cs.Sampler.metropolis_hastings(w, N, t=1, B=B, seed=seed)
    assert np.array_equal(a, oracle.metropolis(w, B, seed, step=1))
    D.close()


def test_config2_filter_1e6_particles_T100(cs, oracle):
    """configs[2] at its size: run() with N = 1e6, T = 100, d = 2 (generateInput()'s model: F = G = I,
    V = W = 0.001 I, src/mcmc.cpp:22-23).  (i) the same seed gives the same 2.8 GB history twice;
    (ii) the history sharded over a device list below the C ABI is bitwise the one-device one;
    (iii) at steps 1, 2, 50 and 99 ALL 1e6 ancestors equal the oracle's chain run over the GPU's own
    weights of the step before, the states are the oracle's proposal from the GPU's own previous states
    and ancestors, and the weights the oracle's reweight_G of the GPU's own states."""
    N, d, T, seed = 1_000_000, 2, 100, 20240
    I = np.eye(d)
    rng = np.random.default_rng(3)
    Y = np.cumsum(np.sqrt(0.001) * rng.standard_normal((d, T)), axis=1) + np.sqrt(0.001) * rng.standard_normal((d, T))
    V = W = 0.001 * I
    args = (N, d, T, Y, np.zeros(d), I, I, I, V, W, 0.0, "metropolis", "mvn")
    one = cs.run(*args, seed=seed, return_ancestors=True)
    X, w, a = one["posterior_x"], one["weights"], one["ancestors"]
    assert X.shape == (T, N, d) and w.shape == (T, N) and a.shape == (T, N)
    assert np.all(np.isfinite(w)) and np.all(np.isfinite(X)) and a.max() < N
    assert np.all(w[0] == 1.0 / N) and not a[0].any()                        # src/mcmc.cpp:85; row 0 unwritten
    again = cs.run(*args, seed=seed, return_ancestors=True)
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], again[k]), k
    del again
    shards = cs.run(*args, seed=seed, return_ancestors=True, devices=[0, 0])
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], shards[k]), k
    del shards
    Qw = oracle.eigen_sqrt(W)
    for t in (1, 2, 50, 99):
        assert np.array_equal(a[t], oracle.metropolis(w[t - 1], 10, seed, step=t)), t
        assert np.allclose(X[t], oracle.propagate(X[t - 1], a[t], I, Qw, "mvn", 0.0, 1.0, seed=seed, step=t),
                           rtol=1e-9, atol=1e-9), t
        assert rel_err(w[t], oracle.reweight(X[t], Y[:, t], I, V, "mvn", 0.0)) < RTOL, t
    # the filter tracks: the weighted mean of the last step sits near the last observation
    est = (w[-1][:, None] * X[-1]).sum(0) / w[-1].sum()
    assert np.all(np.abs(est - Y[:, -1]) < 0.2)


def test_config3_student_t_target_one_gpu_share_of_1e6_chains_1e4_iterations(cs, oracle):
    """configs[3]: MVT (nu = 4) weights of 1e6 particles at d = 64, 1e4 iterations per chain, chains
    sharded 8 ways: rank 3's share (chains 375000 .. 499999, 1.25e9 accept/reject steps over the full
    weight vector) is bit-exact against the oracle's chains; so are the first chains of every other share."""
    import torch
    N, d, nu, B, world = 1_000_000, 64, 4.0, 10_000, 8
    rng = np.random.default_rng(4)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    g = torch.Generator(device="cuda").manual_seed(4)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g) * 1.3
    D = cs.MultiVariateTStudentDistribution(mu, sigma, nu)
    D.ctx.use_torch_stream()
    wd = torch.empty(N, dtype=torch.float64, device="cuda")
    D.pdf_dev(X, wd, log=False)
    torch.cuda.synchronize()
    w = wd.cpu().numpy()
    idx = np.arange(0, N, N // 2048)[:2048]
    assert rel_err(np.log(w[idx]), oracle.logpdf_hoisted(X[idx].cpu().numpy(), mu, sigma, None, "mvt", nu)) < RTOL
    seed, step = 987654321, 1
    count = N // world
    first = 3 * count
    a = torch.empty(count, dtype=torch.int32, device="cuda")
    cs.Sampler.metropolis_hastings_dev(wd, a, B=B, t=step, seed=seed, first=first, ctx=D.ctx)
    torch.cuda.synchronize()
    got = a.cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, oracle.metropolis(w, B, seed, step=step, first=first, count=count))
    for r in range(world):  # a short prefix of every other rank's share
        ar = torch.empty(257, dtype=torch.int32, device="cuda")
        cs.Sampler.metropolis_hastings_dev(wd, ar, B=B, t=step, seed=seed, first=r * count, ctx=D.ctx)
        torch.cuda.synchronize()
        assert np.array_equal(ar.cpu().numpy().astype(np.uint32),
                              oracle.metropolis(w, B, seed, step=step, first=r * count, count=257)), r
    D.close()


def test_config4_d256_one_gpu_share_of_4e6_particles(cs, oracle):
    """configs[4]: d = 256 MVN, 4e6 particles sharded 8 ways: one share (5e5 x 256, 1.0 GB) through
    (i) a 4096-row sample against the oracle, (ii) permutation equivariance, (iii) the exact shift
    identity logp(x; mu) = logp(x - mu; 0), (iv) batch = concatenation of its parts (ragged cut), and
    (v) the share computed as rows [first, first + count) of a larger batch equals the share alone."""
    import torch
    N, d = 500_000, 256
    g = torch.Generator(device="cuda").manual_seed(6)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    rng = np.random.default_rng(5)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    D = cs.MultiVariateNormalDistribution(mu, sigma)
    D.ctx.use_torch_stream()
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D.pdf_dev(X, out)
    torch.cuda.synchronize()
    idx = torch.randint(0, N, (4096,), device="cuda", generator=g)
    assert rel_err(out[idx].cpu().numpy(), oracle.logpdf_hoisted(X[idx].cpu().numpy(), mu, sigma)) < RTOL
    perm = torch.randperm(N, device="cuda", generator=g)
    out2 = torch.empty_like(out)
    Xp = X[perm].contiguous()
    D.pdf_dev(Xp, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out[perm])
    del Xp
    D0 = cs.MultiVariateNormalDistribution(None, sigma)
    Xc = (X - torch.from_numpy(mu).cuda()).contiguous()
    D0.pdf_dev(Xc, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    del Xc
    h = N // 3 + 5
    D.pdf_dev(X[:h], out2[:h])
    D.pdf_dev(X[h:], out2[h:])
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    D.close(); D0.close()
