"""GPU suite: the HIP path (through the C ABI) against the oracle, the committed golden fixtures
and the reference's published known answers.  Run on an MI355X with `pytest -m gpu`.

Tolerances: log-densities 1e-6 relative (the north star's bar; observed ~1e-13 on these
well-conditioned covariances), ancestor indices bit-exact."""
import os

import numpy as np
import pytest

from conftest import DIMS, RESAMPLE_CASES, ROOT, spd

# the dispatch-fuzz tests run this many seeds each (CUSMC_FUZZ_SEEDS=100 for a long soak)
FUZZ_SEEDS = int(os.environ.get("CUSMC_FUZZ_SEEDS", "0"))

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def cs():
    import cusmc_amd
    from cusmc_amd import _lib
    assert _lib.lib().cusmc_device_count() > 0, "no GPU visible: the gpu suite needs an MI355X"
    cusmc_amd.set_seed(2024)
    return cusmc_amd


def rel_err(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


# --- the reference's own published values, through the R-level API ------------------------------

def test_known_answer_mvnpdf(cs):
    v = cs.MVNPDF([0, 0], [0, 0], np.eye(2))  # CuSMC/CuSMC.tex:95-105
    assert abs(v - 0.1591549) < 5e-8 and abs(v - 1 / (2 * np.pi)) < 1e-15


def test_known_answer_mvtpdf(cs):
    v = cs.MVTPDF([0, 0, 0], [0, 0, 0], np.eye(3), 3.0)  # CuSMC/CuSMC.tex:131-142
    assert abs(v - 0.07799708) < 5e-9 and abs(v - 0.0779970835340203) < 1e-14


def test_known_answer_metropolis(cs):
    a = cs.metropolis_hastings([0.0, 0.0], 2, 10)  # man/metropolis_hastings.Rd:22-27
    assert a.dtype == np.float64 and a.tolist() == [0.0, 1.0]


def test_paper_example_shape(cs):
    # CuSMC/CuSMC.tex:150-163: metropolis_hastings(rnorm(100), 100, 10) -> 100 values in 0..99
    w = np.random.default_rng(0).standard_normal(100)
    a = cs.metropolis_hastings(w, 100, 10)
    assert a.shape == (100,) and a.min() >= 0 and a.max() <= 99 and np.all(a == np.floor(a))


# --- densities: golden fixtures (scipy + the reference-faithful restatement) --------------------

@pytest.mark.parametrize("d", DIMS)
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_pdf_against_golden(cs, golden, d, dist):
    g = lambda k: golden["pdf_d%d_%s" % (d, k)]
    nu = float(g("nu"))
    D = (cs.MultiVariateNormalDistribution(g("mu"), g("sigma")) if dist == "mvn"
         else cs.MultiVariateTStudentDistribution(g("mu"), g("sigma"), nu))
    lp = D.pdf_batch(g("X"), g("F"))
    assert rel_err(lp, g(dist + "_logpdf_scipy")) < RTOL
    assert rel_err(lp, np.log(g(dist + "_pdf_oracle"))) < RTOL
    p = D.pdf_batch(g("X"), g("F"), log=False)
    assert rel_err(p, g(dist + "_pdf_oracle")) < RTOL  # the density, what the reference returns
    lw = D.reweight(g("X"), g("y"), g("F"))
    assert rel_err(lw, g(dist + "_reweight_scipy")) < RTOL
    assert rel_err(D.reweight(g("X"), g("y"), g("F"), log=False), g(dist + "_reweight_oracle")) < RTOL
    # scalar interface and getNorm
    assert abs(D.pdf(g("X")[0], g("F")) / g(dist + "_pdf_oracle")[0] - 1) < RTOL
    D.close()


@pytest.mark.parametrize("d", [2, 8, 16, 32, 48, 64, 80, 96, 112, 128, 144, 192, 256])
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_pdf_against_oracle_seeded(cs, oracle, d, dist):
    """Same seeded inputs through the HIP path and the reference-faithful CPU restatement
    (per-particle LU det + inverse), at sizes the oracle finishes in seconds."""
    rng = np.random.default_rng(d * 7 + (dist == "mvt"))
    N = 777 if d <= 64 else 200  # not a multiple of 16: exercises the tail tile
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    X = mu + 1.5 * rng.standard_normal((N, d))
    nu = 4.0
    D = (cs.MultiVariateNormalDistribution(mu, sigma) if dist == "mvn"
         else cs.MultiVariateTStudentDistribution(mu, sigma, nu))
    want = np.log(oracle.pdf_batch(X, mu, sigma, np.eye(d), dist, nu))
    assert rel_err(D.pdf_batch(X), want) < RTOL
    D.close()


@pytest.mark.parametrize("d", [2, 16, 64, 80, 112, 128, 256])
def test_reweight_general_F_against_oracle(cs, oracle, d):
    rng = np.random.default_rng(d)
    N = 500 if d <= 64 else 150
    V = spd(rng, d)
    F = np.eye(d) + 0.2 * rng.standard_normal((d, d)) / np.sqrt(d)
    X = rng.standard_normal((N, d))
    y = rng.standard_normal(d)
    for dist, nu in (("mvn", 0.0), ("mvt", 3.0)):
        D = (cs.MultiVariateNormalDistribution(None, V) if dist == "mvn"
             else cs.MultiVariateTStudentDistribution(None, V, nu))
        want = oracle.reweight(X, y, F, V, dist, nu)
        assert rel_err(D.reweight(X, y, F, log=False), want) < RTOL
        # y changes every time step while F stays: the cached-plan path
        y2 = y + 0.1
        assert rel_err(D.reweight(X, y2, F, log=False), oracle.reweight(X, y2, F, V, dist, nu)) < RTOL
        D.close()


def test_batched_MVNPDF_columns_are_particles(cs, oracle):
    # BASELINE config 1: "MVNPDF() on 1e4 particles, d=8"; x is d x N, columns = particles
    rng = np.random.default_rng(1)
    d, N = 8, 10000
    sigma = spd(rng, d)
    x = rng.standard_normal((d, N))
    p = cs.MVNPDF(x, np.zeros(d), sigma)
    assert p.shape == (N,)
    assert rel_err(p, np.exp(oracle.logpdf_hoisted(x.T, None, sigma))) < RTOL
    assert rel_err(p[:64], oracle.pdf_batch(x.T[:64], np.zeros(d), sigma, np.eye(d))) < RTOL


def test_edge_cases(cs):
    D = cs.MultiVariateNormalDistribution(np.zeros(64), np.eye(64))
    assert D.pdf_batch(np.zeros((0, 64))).shape == (0,)            # empty batch
    one = D.pdf_batch(np.zeros((1, 64)))                           # a single particle (ragged tile)
    assert abs(one[0] + 32 * np.log(2 * np.pi)) < 1e-10
    far = D.pdf_batch(np.full((3, 64), 1e3))                       # density underflows, log does not
    assert np.all(np.isfinite(far)) and np.all(D.pdf_batch(np.full((3, 64), 1e3), log=False) == 0.0)
    assert np.isnan(D.pdf_batch(np.full((1, 64), np.nan))[0])     # NaN in, NaN out
    D.close()
    with pytest.raises(cs.CusmcError) as e:                        # not SPD is reported, not NaN
        cs.MultiVariateNormalDistribution([0, 0], [[1.0, 2.0], [2.0, 1.0]])
    assert e.value.code == 2
    with pytest.raises(cs.CusmcError):
        cs.MultiVariateTStudentDistribution([0, 0], np.eye(2), -1.0)


def test_strided_and_unaligned_batches(cs, oracle):
    """ldx > d and odd strides take the generic kernel; results must not depend on the route."""
    import torch
    rng = np.random.default_rng(9)
    d, N = 64, 300
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    want = oracle.logpdf_hoisted(rng.standard_normal((1, d)), mu, sigma)  # warm the oracle
    Xh = rng.standard_normal((N, d + 3))
    want = oracle.logpdf_hoisted(np.ascontiguousarray(Xh[:, :d]), mu, sigma)
    D = cs.MultiVariateNormalDistribution(mu, sigma)
    Xd = torch.from_numpy(Xh).cuda()
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D.ctx.use_torch_stream()
    D.pdf_dev(Xd[:, :d], out)                     # ldx = 67: odd -> generic kernel
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), want) < RTOL
    Xe = torch.from_numpy(np.ascontiguousarray(Xh[:, :d + 2])).cuda()
    D.pdf_dev(Xe[:, :d], out)                     # ldx = 66: even -> MFMA kernel with a row gap
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), want) < RTOL
    D.close()


@pytest.mark.parametrize("N,d", [(1_000_000, 64), (250_000, 128), (100_003, 256)])
def test_full_size_properties(cs, oracle, N, d):
    """BASELINE's headline size, checked through size-independent properties: (i) a 4096-row
    sample against the oracle, (ii) permutation equivariance, (iii) the exact shift identity
    logp(x; mu) = logp(x - mu; 0), (iv) the batch equals the concatenation of its halves."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    rng = np.random.default_rng(1)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    D = cs.MultiVariateNormalDistribution(mu, sigma)
    D.ctx.use_torch_stream()
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D.pdf_dev(X, out)
    torch.cuda.synchronize()
    idx = torch.randint(0, N, (4096,), device="cuda", generator=g)
    want = oracle.logpdf_hoisted(X[idx].cpu().numpy(), mu, sigma)
    assert rel_err(out[idx].cpu().numpy(), want) < RTOL
    perm = torch.randperm(N, device="cuda", generator=g)
    out2 = torch.empty_like(out)
    D.pdf_dev(X[perm].contiguous(), out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out[perm])           # bitwise: no cross-particle coupling
    D0 = cs.MultiVariateNormalDistribution(None, sigma)
    out3 = torch.empty_like(out)
    D0.pdf_dev((X - torch.from_numpy(mu).cuda()).contiguous(), out3)
    torch.cuda.synchronize()
    assert torch.equal(out3, out)
    h = N // 2 + 3
    D.pdf_dev(X[:h], out2[:h])
    D.pdf_dev(X[h:], out2[h:])
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    D.close(); D0.close()


@pytest.mark.parametrize("N,ldx", [(65_536 + 5, 64), (125_000, 64), (262_144 + 5, 64), (400_001, 66), (1_000_000, 80), (999_999, 64)])
def test_assembly_kernel_against_oracle_and_compiled_kernel(cs, oracle, N, ldx):
    """The hand-written assembly kernel (kernels/logpdf_nb4_gfx950.s: d = 64, zero mean, MVN log-density, N >= 65536):
    EVERY row against the hoisted oracle, strided rows with NaN in the padding, a ragged last tile -- and bitwise
    against the compiled kernel, which serves the same rows when they arrive in pieces below the assembly kernel's
    threshold (its tail pool hands tiles to whichever wave draws the ticket: the arithmetic must not notice).
    Repeated launches reuse the alternating pool-counter blocks."""
    import torch
    d = 64
    rng = np.random.default_rng(N)
    sigma = spd(rng, d)
    g = torch.Generator(device="cuda").manual_seed(N)
    buf = torch.full((N, ldx), float("nan"), dtype=torch.float64, device="cuda")
    buf[:, :d] = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    X = buf[:, :d]
    D = cs.MultiVariateNormalDistribution(None, sigma)
    D.ctx.use_torch_stream()
    out = torch.full((N + 16,), -7.0, dtype=torch.float64, device="cuda")
    for _ in range(3):
        out[:N].fill_(float("nan"))
        D.pdf_dev(X, out[:N])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.all(got[N:] == -7.0)
    want = oracle.logpdf_hoisted(X.cpu().numpy(), None, sigma)
    assert rel_err(got[:N], want) < RTOL
    pieces = torch.empty(N, dtype=torch.float64, device="cuda")
    step = 60_000  # (< 16 rounds of 256 tiles: the compiled kernel)
    for lo in range(0, N, step):
        hi = min(N, lo + step)
        D.pdf_dev(X[lo:hi], pieces[lo:hi])
    torch.cuda.synchronize()
    assert torch.equal(pieces, out[:N])
    D.close()


@pytest.mark.parametrize("d,N", [(64, 1), (64, 15), (64, 16), (64, 17), (64, 256 * 16 - 1), (64, 256 * 16), (64, 256 * 16 + 1),
                                 (64, 3 * 256 * 16 - 5), (64, 8 * 256 * 16 + 7), (32, 256 * 16 * 2 + 33), (16, 70_001),
                                 (48, 12_289), (96, 256 * 16 + 9)])
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_tile_deal_boundaries(cs, oracle, d, N, dist):
    """Every particle is evaluated exactly once whatever N is relative to the deal (16-particle
    tiles, one workgroup per CU, rounds of G tiles): ALL outputs against the hoisted-factor oracle,
    on a buffer pre-filled with NaN and guarded by a sentinel past the end."""
    import torch
    rng = np.random.default_rng(N + d)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    Xh = mu + rng.standard_normal((N, d))
    X = torch.from_numpy(Xh).cuda()
    out = torch.full((N + 64,), float("nan"), dtype=torch.float64, device="cuda")
    D = (cs.MultiVariateNormalDistribution(mu, sigma) if dist == "mvn"
         else cs.MultiVariateTStudentDistribution(mu, sigma, 4.0))
    D.ctx.use_torch_stream()
    D.pdf_dev(X, out[:N])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.all(np.isnan(got[N:])), "wrote past the end of the output"
    want = oracle.logpdf_hoisted(Xh, mu, sigma, None, dist, 4.0)
    assert rel_err(got[:N], want) < RTOL
    D.close()


@pytest.mark.parametrize("d", [17, 24, 31, 40, 65, 81, 100, 127, 129, 144, 150, 160, 170, 176, 191, 192, 200, 208, 220, 224, 239, 240, 255, 256])
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_padded_dimensions(cs, oracle, d, dist):
    """d not a multiple of 16 runs on the matrix cores with the factor zero-padded: every output
    against the hoisted-factor oracle (odd N: short last tile; odd d: rows only 8-byte aligned),
    the general-F reweight form too, and no leakage between neighbouring rows -- a row of NaN / Inf
    makes exactly its own output NaN, whatever sits in the rows around it."""
    import torch
    rng = np.random.default_rng(3 * d)
    N = 16 * 256 * 2 + 11
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    Xh = mu + rng.standard_normal((N, d))
    bad = [0, 5, 777, N - 1]
    Xh[bad[0]] = np.nan
    Xh[bad[1], 0] = np.inf       # first column of a row: what the previous row's padded block would touch
    Xh[bad[2], d - 1] = np.nan
    Xh[bad[3]] = np.inf
    X = torch.from_numpy(Xh).cuda()
    out = torch.full((N + 8,), -1.0, dtype=torch.float64, device="cuda")
    D = (cs.MultiVariateNormalDistribution(mu, sigma) if dist == "mvn"
         else cs.MultiVariateTStudentDistribution(mu, sigma, 4.0))
    D.ctx.use_torch_stream()
    D.pdf_dev(X, out[:N])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.all(got[N:] == -1.0)
    good = np.ones(N, bool); good[bad] = False
    assert not np.isfinite(got[:N][~good]).any()
    assert np.isfinite(got[:N][good]).all()
    want = oracle.logpdf_hoisted(Xh[good], mu, sigma, None, dist, 4.0)
    assert rel_err(got[:N][good], want) < RTOL
    # reweight_G with a general F (QL-rotated affine plan, padded likewise)
    F = np.eye(d) + 0.05 * rng.standard_normal((d, d))
    y = rng.standard_normal(d)
    Xg = torch.from_numpy(Xh[good][:3000]).cuda().contiguous()
    w = torch.empty(Xg.shape[0], dtype=torch.float64, device="cuda")
    D.reweight_dev(Xg, y, F, w, log=True)
    torch.cuda.synchronize()
    R = y[None, :] - Xh[good][:3000] @ F.T
    want2 = oracle.logpdf_hoisted(R, None, sigma, None, dist, 4.0)
    assert rel_err(w.cpu().numpy(), want2) < RTOL
    D.close()


@pytest.mark.parametrize("d", [257, 300, 320, 384])
def test_beyond_256_dimensions(cs, oracle, d):
    """256 < d <= CUSMC_MAX_DIM = 384 (the reference has no limit below tgamma's overflow at
    d ~ 340, src/statistics.cc.cpp:302) is served by the shape-agnostic kernels: log-densities (centred with mu,
    reweight with a dense F; Normal and Student-t) against the hoisted AND the reference-faithful oracle, proposal
    draws and a propagate step against the oracle, a small filter run bit-exact in its ancestors."""
    import torch
    rng = np.random.default_rng(d)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    N = 333
    Xh = mu + rng.standard_normal((N, d))
    X = torch.from_numpy(Xh).cuda()
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    F = np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d)
    y = rng.standard_normal(d)
    for dist, nu in (("mvn", 0.0), ("mvt", 4.0)):
        D = (cs.MultiVariateNormalDistribution(mu, sigma) if dist == "mvn" else cs.MultiVariateTStudentDistribution(mu, sigma, nu))
        D.ctx.use_torch_stream()
        D.pdf_dev(X, out)
        torch.cuda.synchronize()
        assert rel_err(out.cpu().numpy(), oracle.logpdf_hoisted(Xh, mu, sigma, None, dist, nu)) < RTOL
        faithful = oracle.pdf_batch(Xh[:6], mu, sigma, np.eye(d), dist, nu)
        if np.all(faithful > 0) and np.all(np.isfinite(faithful)):  # (the reference's density form under/overflows up here)
            assert rel_err(out[:6].cpu().numpy(), np.log(faithful)) < RTOL
        D.reweight_dev(X, y, F, out, log=True)
        torch.cuda.synchronize()
        assert rel_err(out.cpu().numpy(), oracle.logpdf_hoisted(y[None, :] - Xh @ F.T, None, sigma, None, dist, nu)) < RTOL
        assert rel_err(D.pdf_batch(Xh), oracle.logpdf_hoisted(Xh, mu, sigma, None, dist, nu)) < RTOL   # host-pointer entry
        Q = oracle.eigen_sqrt(0.2 * spd(rng, d))
        got = D.sample(Q, 200, count=50, seed=42, step=6)
        want, _ = oracle.initialize(50, mu, Q, dist, nu, 1.0, seed=42, step=6)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9)
        G = 0.9 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d)
        a = rng.integers(0, N, N).astype(np.uint32)
        Xo = torch.empty(N, d, dtype=torch.float64, device="cuda")
        cs.api.propagate_dev(X, torch.from_numpy(a.astype(np.int32)).cuda(), G, Q, Xo, dist, nu, 1.0, seed=7, step=3, ctx=D.ctx)
        torch.cuda.synchronize()
        assert np.allclose(Xo.cpu().numpy(), oracle.propagate(Xh, a, G, Q, dist, nu, 1.0, seed=7, step=3), rtol=1e-9, atol=1e-9)
        D.close()
    T, Np = 3, 64
    Y = np.cumsum(0.1 * rng.standard_normal((d, T)), axis=1)
    I = np.eye(d)
    res = cs.run(Np, d, T, Y, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=5, return_ancestors=True)
    Xr, wr, ar = oracle.pf_run(Y.T, Np, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, "mvn", 0.0, B=10, seed=5, hoisted=True)
    assert np.array_equal(res["ancestors"], ar) and np.allclose(res["posterior_x"], Xr, atol=1e-9)
    with pytest.raises(cs.CusmcError) as e:
        cs.MultiVariateNormalDistribution(np.zeros(385), np.eye(385))
    from cusmc_amd import _lib
    assert e.value.code == _lib.ERANGE


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 6))
def test_logpdf_random_shapes(cs, oracle, seed):
    """Dispatch fuzz: random d in [1, 256] (CUSMC_MAX_DIM), random N, row stride, base alignment, distribution and
    form (centred with mu, or reweight with F = I / general F), every output against the hoisted
    oracle.  Whatever kernel the library picks, the numbers are the reference's."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        d = int(rng.choice([rng.integers(1, 17), rng.integers(17, 129), rng.integers(129, 257)]))
        N = int(rng.integers(1, 3000))
        ldx = d + int(rng.choice([0, 0, 1, 2, 7]))
        off = int(rng.choice([0, 1]))
        dist = str(rng.choice(["mvn", "mvt"]))
        nu = float(rng.choice([3.0, 4.0, 2.5, 30.0, 9.0]))
        form = str(rng.choice(["pdf", "reweight_I", "reweight_F"]))
        sigma, mu = spd(rng, d), rng.standard_normal(d)
        Xh = rng.standard_normal((N, d))
        buf = torch.full((N * ldx + 2,), float("nan"), dtype=torch.float64, device="cuda")
        X = buf[off:off + N * ldx].view(N, ldx)[:, :d]
        X.copy_(torch.from_numpy(Xh))
        out = torch.full((N + 2,), -7.0, dtype=torch.float64, device="cuda")
        D = (cs.MultiVariateNormalDistribution(mu, sigma) if dist == "mvn"
             else cs.MultiVariateTStudentDistribution(mu, sigma, nu))
        D.ctx.use_torch_stream()
        if form == "pdf":
            D.pdf_dev(X, out[:N])
            want = oracle.logpdf_hoisted(Xh, mu, sigma, None, dist, nu)
        else:
            F = np.eye(d) if form == "reweight_I" else np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d)
            y = rng.standard_normal(d)
            D.reweight_dev(X, y, F, out[:N], log=True)
            want = oracle.logpdf_hoisted(y[None, :] - Xh @ F.T, None, sigma, None, dist, nu)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        tag = (seed, case, d, N, ldx, off, dist, form)
        assert np.all(got[N:] == -7.0), tag
        assert rel_err(got[:N], want) < RTOL, tag
        D.close()


def test_full_size_properties_student_t(cs, oracle):
    """BASELINE configs[3] shape (nu = 4, 1e6 x d = 64) through the same size-independent properties."""
    import torch
    N, d, nu = 1_000_000, 64, 4.0
    g = torch.Generator(device="cuda").manual_seed(9)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    rng = np.random.default_rng(3)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    D = cs.MultiVariateTStudentDistribution(mu, sigma, nu)
    D.ctx.use_torch_stream()
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D.pdf_dev(X, out)
    torch.cuda.synchronize()
    idx = torch.randint(0, N, (4096,), device="cuda", generator=g)
    want = oracle.logpdf_hoisted(X[idx].cpu().numpy(), mu, sigma, None, "mvt", nu)
    assert rel_err(out[idx].cpu().numpy(), want) < RTOL
    perm = torch.randperm(N, device="cuda", generator=g)
    out2 = torch.empty_like(out)
    D.pdf_dev(X[perm].contiguous(), out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out[perm])
    # the Student-t and the Normal density share q: logp_t = c_t - (nu+d)/2 log1p(q/nu), q = -2 (logp_n - c_n)
    Dn = cs.MultiVariateNormalDistribution(mu, sigma)
    outn = torch.empty_like(out)
    Dn.pdf_dev(X, outn)
    torch.cuda.synchronize()
    q = -2.0 * (outn - Dn.lognorm())
    back = D.lognorm() - 0.5 * (nu + d) * torch.log1p(q / nu)
    assert float(((back - out).abs() / out.abs()).max()) < 1e-12
    D.close(); Dn.close()


# --- resampler: bit-exact index sequences -----------------------------------------------------------

@pytest.mark.parametrize("name", RESAMPLE_CASES)
@pytest.mark.parametrize("B", [1, 10, 37])
def test_resampler_bit_exact_golden(cs, golden, name, B):
    w = golden["resample_%s_w" % name]
    a = cs.Sampler.metropolis_hastings(w, None, t=1, B=B, seed=20240 + B)
    assert a.dtype == np.uint32
    assert np.array_equal(a, golden["resample_%s_B%d" % (name, B)])


@pytest.mark.parametrize("N,B", [(1, 5), (63, 3), (100_000, 10), (100_000, 101)])
def test_resampler_bit_exact_oracle(cs, oracle, N, B):
    # BASELINE config 2 shape: weights are d=32 MVN densities (~1e-20: the ratio form matters)
    rng = np.random.default_rng(N + B)
    d = 32
    sigma = spd(rng, d)
    X = rng.standard_normal((N, d)) @ np.linalg.cholesky(sigma).T
    w = np.exp(oracle.logpdf_hoisted(X, None, sigma))
    seed = 0x1234_5678_9ABC_DEF0
    a = cs.Sampler.metropolis_hastings(w, N, t=7, B=B, seed=seed)
    assert np.array_equal(a, oracle.metropolis(w, B, seed, step=7))


@pytest.mark.parametrize("B,seed", [(2, 1), (10, 99), (37, 12345678901234567), (100, 7)])
@pytest.mark.parametrize("kind", ["densities", "adversarial"])
def test_resampler_large_N_truncated_table(cs, oracle, kind, B, seed):
    """N = 1e6 (the weight vector no longer fits one XCD's L2): the chain gathers from the table of
    high words and settles undecidable steps with the exact test -- ancestors must still be
    bit-identical to the oracle's plain chain.  'adversarial' plants what the shortcut must not
    decide by itself: runs of equal weights (ratio exactly 1), ratios a hair from every u-independent
    boundary, zeros, denormals, huge values, negatives, NaN."""
    N = 1_000_000 if B == 10 else 131_072 if B == 100 else 450_000  # (B = 100, N = 131072: the kernel with the table's head in LDS)
    rng = np.random.default_rng(42 + B)
    if kind == "densities":
        w = np.exp(-0.5 * rng.chisquare(32, N)) * 1e-20
    else:
        w = rng.random(N)
        w[::7] = 0.5                                   # ties
        w[1::7] = 0.5 * (1 + rng.integers(-4, 5, len(w[1::7])) * 2.0 ** -21)   # inside the truncation error
        w[2::1001] = 0.0
        w[3::1001] = 5e-324
        w[4::1001] = 1e-310
        w[5::1001] = 1e300
        w[6::1001] = -0.25
        w[7::5003] = np.nan
        w[8::5003] = np.inf
    a = cs.Sampler.metropolis_hastings(w, N, t=3, B=B, seed=seed)
    assert np.array_equal(a, oracle.metropolis(w, B, seed, step=3))


@pytest.mark.parametrize("N,B", [(1, 5), (63, 3), (100_000, 10), (100_000, 101), (700_000, 10)])
def test_log_weight_resampler_bit_exact(cs, oracle, N, B):
    """cusmc_metropolis_log: accept iff u <= exp(lw[j] - lw[k]), with exp as ONE fixed sequence of
    rounded operations mirrored in the oracle -- so the index sequences are bit-identical, including
    far below the density form's underflow, with -inf (zero) weights, NaN, and shard by shard."""
    import torch
    rng = np.random.default_rng(N + B)
    lw = -0.5 * rng.chisquare(64, N) - 1500.0
    if N > 100:
        lw[::13] = -np.inf
        lw[5::1001] = np.nan
        lw[7::1001] = 0.0
    seed = 0xFEED_FACE_CAFE_BEEF
    a = cs.Sampler.metropolis_hastings_log(lw, N, t=4, B=B, seed=seed)
    want = oracle.metropolis_log(lw, B, seed, step=4)
    assert np.array_equal(a, want)
    if N >= 1000:
        lwd = torch.from_numpy(lw).cuda()
        first, count = N // 3, N // 2
        part = torch.empty(count, dtype=torch.int32, device="cuda")
        cs.Sampler.metropolis_hastings_log_dev(lwd, part, B=B, t=4, seed=seed, first=first)
        torch.cuda.synchronize()
        assert np.array_equal(part.cpu().numpy().astype(np.uint32), want[first:first + count])


def test_resampler_contract3_rare_paths(cs, oracle):
    """RNG contract 3's two completions, forced (a random run meets them once in 2^32 resp. 2^12 steps):
    (1) the ratio INSIDE the 32-bit cell of u -- w[j] / w[k] = (a + 1/2) 2^-32 for the very a the step draws -- so that
        the next 53 bits decide: the plain chain (N = 2), the truncated-table chain (N = 1e6, where the undecided step
        falls through to the exact test) and the log-weight chain;
    (2) index candidates redrawn by Lemire's rejection: N = 3 * 2^20 redraws one in 4096.
    All bit-identical to the oracle (whose own exact-arithmetic restatement is tests/test_oracle.py)."""
    hits = 0
    for seed in range(40):
        step = 2
        r = oracle.philox4x32_10([0, 0, step, 1], [seed, 0])
        a, j = int(r[0]), (int(r[1]) * 2) >> 32
        if j != 1:
            continue
        w = np.array([1.0, (a + 0.5) * 2.0 ** -32])
        assert np.array_equal(cs.Sampler.metropolis_hastings(w, 2, t=step, B=1, seed=seed), oracle.metropolis(w, 1, seed, step))
        lw = np.log(w)
        assert np.array_equal(cs.Sampler.metropolis_hastings_log(lw, 2, t=step, B=1, seed=seed), oracle.metropolis_log(lw, 1, seed, step))
        hits += 1
    assert hits >= 5
    # the truncated-table chain: chain 0's first proposal meets a weight inside its cell
    N, B, seed, step = 1_000_000, 2, 99, 3
    r = oracle.philox4x32_10([0, 0, step, 1], [seed, 0])
    a, j = int(r[0]), (int(r[1]) * N) >> 32
    w = np.ones(N)
    if j != 0:
        w[j] = (a + 0.5) * 2.0 ** -32
    assert np.array_equal(cs.Sampler.metropolis_hastings(w, N, t=step, B=B, seed=seed), oracle.metropolis(w, B, seed, step))
    # redraws
    N = 3 * 2 ** 20
    w = np.random.default_rng(8).random(N)
    for B in (1, 10):
        assert np.array_equal(cs.Sampler.metropolis_hastings(w, N, t=4, B=B, seed=31337), oracle.metropolis(w, B, 31337, step=4))
    lw = np.log(w)
    assert np.array_equal(cs.Sampler.metropolis_hastings_log(lw, N, t=4, B=5, seed=31337), oracle.metropolis_log(lw, 5, 31337, step=4))


def test_resampler_shards_compose(cs, oracle):
    """Chains [first, first+count) computed separately equal the single launch (global Philox
    indices): what the multi-GPU path relies on."""
    import torch
    rng = np.random.default_rng(3)
    N, B = 5003, 10
    w = rng.random(N)
    wd = torch.from_numpy(w).cuda()
    ctx = cs.api.default_context().use_torch_stream()
    parts = []
    from cusmc_amd.sharding import shard_range
    for r in range(3):
        first, count = shard_range(N, r, 3)
        a = torch.empty(count, dtype=torch.int32, device="cuda")
        cs.Sampler.metropolis_hastings_dev(wd, a, B=B, t=2, seed=11, first=first, ctx=ctx)
        parts.append(a)
    torch.cuda.synchronize()
    got = torch.cat(parts).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, oracle.metropolis(w, B, 11, step=2))


def test_resampler_full_size_properties(cs):
    """N = 1e6, B = 10 (BASELINE config 3's per-step shape): range, determinism, and that the
    chain targets w (frequency of the heavy class)."""
    import torch
    N = 1_000_000
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    w[::10] = 9.0                                   # 10% of particles carry 50% of the mass
    a = torch.empty(N, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    ctx = cs.api.default_context().use_torch_stream()
    cs.Sampler.metropolis_hastings_dev(w, a, B=30, t=1, seed=5, ctx=ctx)
    cs.Sampler.metropolis_hastings_dev(w, b, B=30, t=1, seed=5, ctx=ctx)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) < N
    heavy = float((a % 10 == 0).double().mean())
    assert abs(heavy - 0.5) < 0.02


# --- draws and the filter ---------------------------------------------------------------------------

@pytest.mark.parametrize("d", [2, 5, 8, 16, 33, 64, 81, 100, 128, 160, 256])
@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 5.0), ("mvt", 1.5), ("mvt", 4.0), ("mvt", 2.0), ("mvt", 1.0), ("mvt", 7.0),
                                     ("mvt", 16.0), ("mvt", 17.0)])
def test_draws_match_oracle(cs, oracle, d, dist, nu):
    """Same Philox counters, same transform: the draws agree to rounding (libm vs ocml log/
    sin/cos differ in the last bits, so this is a tolerance, not bit-exact)."""
    rng = np.random.default_rng(d)
    S = spd(rng, d)
    Q = oracle.eigen_sqrt(S)
    mu = rng.standard_normal(d)
    D = (cs.MultiVariateNormalDistribution(mu, S) if dist == "mvn"
         else cs.MultiVariateTStudentDistribution(mu, S, nu))
    for scale, compat in ((1.0, False), (3 ** 0.5, True)):
        got = D.sample(Q, 200, count=300, compat=compat, seed=42, step=6)
        want, _ = oracle.initialize(300, mu, Q, dist, nu, scale, seed=42, step=6)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9)
    D.close()


@pytest.mark.parametrize("d", [2, 8, 16, 17, 32, 40, 48, 64, 65, 80, 96, 100, 113, 128, 129, 144, 150, 192, 201, 256])
@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 4.0), ("mvt", 3.0), ("mvt", 2.5)])
def test_propagate_matches_oracle(cs, oracle, d, dist, nu):
    """propagate_K (src/mcmc.cpp:112-140): gather by ancestor + G x + Q xi, device-resident, against the
    oracle on the same Philox counters; 16 <= d <= 128 take the MFMA kernel (padded when d is not a
    multiple of 16, two launches above d = 96), the others the generic one.
    Also the shard identity the multi-GPU path relies on: rows [first, first+count) computed alone
    equal the same rows of the full launch."""
    import torch
    rng = np.random.default_rng(d + 17)
    N = 1000 + 7  # not a multiple of 16
    Xp = rng.standard_normal((N, d))
    a = rng.integers(0, N, N).astype(np.uint32)
    G = 0.9 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d)   # dense, asymmetric
    Q = oracle.eigen_sqrt(0.2 * spd(rng, d))
    want = oracle.propagate(Xp, a, G, Q, dist, nu, 1.0, seed=77, step=5)
    ctx = cs.api.default_context().use_torch_stream()
    Xd = torch.from_numpy(Xp).cuda()
    ad = torch.from_numpy(a.astype(np.int32)).cuda()
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd, ad, G, Q, out, dist, nu, 1.0, seed=77, step=5, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(out.cpu().numpy(), want, rtol=1e-9, atol=1e-9)
    first, count = 333, 401
    part = torch.empty(count, d, dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd, ad[first:first + count].contiguous(), G, Q, part, dist, nu, 1.0, seed=77, step=5,
                         first=first, ctx=ctx)
    torch.cuda.synchronize()
    assert torch.equal(part, out[first:first + count])


@pytest.mark.parametrize("d", [16, 17, 31, 32, 48, 64, 65, 96, 100, 112, 113, 128, 129, 144, 150, 177, 192, 201, 208, 224, 240, 255, 256])
@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 4.0), ("mvt", 3.0), ("mvt", 2.5)])
def test_propagate_with_a_triangular_factor(cs, oracle, d, dist, nu):
    """A LOWER TRIANGULAR Q (the Cholesky factor cusmc_pf_run_* passes down; recognised by its zero upper part)
    takes the kernels' triangular instantiations -- Q xi over the k-blocks kb <= cb only, 97 <= d <= 112 in one
    launch for the Normal proposal: against the oracle's dense loops on the same Q, with a dense G, a diagonal G
    and for initialize() (m0 instead of G), shards composing as for the dense kernels."""
    import torch
    rng = np.random.default_rng(d + 1700)
    N = 1000 + 7
    Xp = rng.standard_normal((N, d))
    a = rng.integers(0, N, N).astype(np.uint32)
    Q = np.linalg.cholesky(0.2 * spd(rng, d))
    assert np.all(np.triu(Q, 1) == 0.0)
    ctx = cs.api.default_context().use_torch_stream()
    Xd = torch.from_numpy(Xp).cuda()
    ad = torch.from_numpy(a.astype(np.int32)).cuda()
    for G in (0.9 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d), np.diag(0.5 + rng.random(d))):
        want = oracle.propagate(Xp, a, G, Q, dist, nu, 1.0, seed=177, step=5)
        out = torch.full((N + 1, d), float("nan"), dtype=torch.float64, device="cuda")
        cs.api.propagate_dev(Xd, ad, G, Q, out[:N], dist, nu, 1.0, seed=177, step=5, ctx=ctx)
        torch.cuda.synchronize()
        assert np.allclose(out[:N].cpu().numpy(), want, rtol=1e-9, atol=1e-9)
        assert bool(torch.isnan(out[N]).all())
        first, count = 333, 401
        part = torch.empty(count, d, dtype=torch.float64, device="cuda")
        cs.api.propagate_dev(Xd, ad[first:first + count].contiguous(), G, Q, part, dist, nu, 1.0, seed=177, step=5,
                             first=first, ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(part, out[first:first + count])
    m0 = rng.standard_normal(d)
    want0, _ = oracle.initialize(N, m0, Q, dist, nu, 1.0, seed=177, step=0)
    init = torch.empty(N, d, dtype=torch.float64, device="cuda")
    cs.api.initialize_dev(m0, Q, init, dist, nu, 1.0, seed=177, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(init.cpu().numpy(), want0, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("d", [1, 2, 9, 16, 64, 70, 200])
@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 4.0), ("mvt", 2.5), ("mvt", 2.0), ("mvt", 5.0), ("mvt", 12.0)])
def test_propagate_diagonal_models(cs, oracle, d, dist, nu):
    """Diagonal G and Q (random-walk / independent-component models, e.g. generateInput(),
    src/mcmc.cpp:22-23) take the lane-per-component-pair kernel at any d: against the oracle,
    which runs the dense loops, and shard-composable like the general kernels."""
    import torch
    rng = np.random.default_rng(d + 170)
    N = 2000 + 3
    Xp = rng.standard_normal((N, d))
    a = rng.integers(0, N, N).astype(np.uint32)
    G = np.diag(0.5 + rng.random(d))
    Q = np.diag(0.1 + rng.random(d))
    want = oracle.propagate(Xp, a, G, Q, dist, nu, 1.0, seed=78, step=4)
    ctx = cs.api.default_context().use_torch_stream()
    Xd = torch.from_numpy(Xp).cuda()
    ad = torch.from_numpy(a.astype(np.int32)).cuda()
    out = torch.full((N + 1, d), float("nan"), dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd, ad, G, Q, out[:N], dist, nu, 1.0, seed=78, step=4, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(out[:N].cpu().numpy(), want, rtol=1e-9, atol=1e-9)
    assert bool(torch.isnan(out[N]).all())
    first, count = 777, 1001
    part = torch.empty(count, d, dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd, ad[first:first + count].contiguous(), G, Q, part, dist, nu, 1.0, seed=78, step=4,
                         first=first, ctx=ctx)
    torch.cuda.synchronize()
    assert torch.equal(part, out[first:first + count])
    # initialize(): no G, m0 instead
    m0 = rng.standard_normal(d)
    want0, _ = oracle.initialize(N, m0, Q, dist, nu, 1.0, seed=78, step=0)
    init = torch.empty(N, d, dtype=torch.float64, device="cuda")
    cs.api.initialize_dev(m0, Q, init, dist, nu, 1.0, seed=78, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(init.cpu().numpy(), want0, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("d", [2, 64])
def test_propagate_diagonal_models_on_8_byte_aligned_rows(cs, oracle, d):
    """The diagonal kernel moves 16-byte pieces when d is even and both batches are 16-byte aligned;
    a batch that starts 8 bytes into an allocation takes the element-wise path: same values."""
    import torch
    rng = np.random.default_rng(d + 171)
    N = 1500 + 1
    Xp = rng.standard_normal((N, d))
    a = rng.integers(0, N, N).astype(np.uint32)
    G = np.diag(0.5 + rng.random(d))
    Q = np.diag(0.1 + rng.random(d))
    want = oracle.propagate(Xp, a, G, Q, "mvn", 0.0, 1.0, seed=79, step=3)
    ctx = cs.api.default_context().use_torch_stream()
    ad = torch.from_numpy(a.astype(np.int32)).cuda()
    flat_in = torch.empty(N * d + 1, dtype=torch.float64, device="cuda")
    flat_out = torch.full((N * d + 2,), float("nan"), dtype=torch.float64, device="cuda")
    Xd = flat_in[1:].view(N, d)
    Xd.copy_(torch.from_numpy(Xp))
    out = flat_out[1:N * d + 1].view(N, d)
    assert Xd.data_ptr() % 16 == 8 and out.data_ptr() % 16 == 8
    cs.api.propagate_dev(Xd, ad, G, Q, out, "mvn", 0.0, 1.0, seed=79, step=3, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(out.cpu().numpy(), want, rtol=1e-9, atol=1e-9)
    assert bool(torch.isnan(flat_out[0])) and bool(torch.isnan(flat_out[-1]))
    aligned = torch.empty(N, d, dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd.clone(), ad, G, Q, aligned, "mvn", 0.0, 1.0, seed=79, step=3, ctx=ctx)
    torch.cuda.synchronize()
    assert torch.equal(aligned, out)


@pytest.mark.parametrize("d", [16, 24, 64, 100, 128, 131, 160, 256])
@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 4.0)])
def test_propagate_diagonal_G_dense_Q(cs, oracle, d, dist, nu):
    """A diagonal G (random walk / AR(1) per component) under a dense Q: the matrix-core kernel runs
    ONE product and gathers the ancestor's row straight into its output layout -- same values as the
    dense-G form of the same kernel, against the oracle's dense loops, shard-composable, on and off the
    16-grid, and without ancestors (a == NULL: a[i] = i)."""
    import torch
    rng = np.random.default_rng(d + 270)
    N = 1500 + 7
    Xp = rng.standard_normal((N, d))
    a = rng.integers(0, N, N).astype(np.uint32)
    G = np.diag(0.5 + rng.random(d))
    Q = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    want = oracle.propagate(Xp, a, G, Q, dist, nu, 1.0, seed=79, step=5)
    ctx = cs.api.default_context().use_torch_stream()
    Xd = torch.from_numpy(Xp).cuda()
    ad = torch.from_numpy(a.astype(np.int32)).cuda()
    out = torch.full((N + 1, d), float("nan"), dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd, ad, G, Q, out[:N], dist, nu, 1.0, seed=79, step=5, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(out[:N].cpu().numpy(), want, rtol=1e-9, atol=1e-9)
    assert bool(torch.isnan(out[N]).all())
    first, count = 333, 1001
    part = torch.empty(count, d, dtype=torch.float64, device="cuda")
    cs.api.propagate_dev(Xd, ad[first:first + count].contiguous(), G, Q, part, dist, nu, 1.0, seed=79, step=5,
                         first=first, ctx=ctx)
    torch.cuda.synchronize()
    assert torch.equal(part, out[first:first + count])
    ident = np.arange(N, dtype=np.uint32)
    cs.api.propagate_dev(Xd, None, G, Q, out[:N], dist, nu, 1.0, seed=79, step=5, ctx=ctx)
    torch.cuda.synchronize()
    assert np.allclose(out[:N].cpu().numpy(), oracle.propagate(Xp, ident, G, Q, dist, nu, 1.0, seed=79, step=5),
                       rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 4))
def test_propagate_random_shapes(cs, oracle, seed):
    """Dispatch fuzz for the proposal draws: random d in [1, 256], dense or diagonal G and Q, MVN or
    Student-t, random shard, with and without ancestors, against the oracle's dense loops."""
    import torch
    rng = np.random.default_rng(2000 + seed)
    ctx = cs.api.default_context().use_torch_stream()
    for case in range(8):
        d = int(rng.choice([rng.integers(1, 17), rng.integers(17, 129), rng.integers(129, 257)]))
        N = int(rng.integers(1, 1500))
        dist = str(rng.choice(["mvn", "mvt"]))
        nu = float(rng.choice([3.0, 4.0, 1.5, 6.0, 2.5]))
        diag = bool(rng.integers(0, 2))
        G = np.diag(0.5 + rng.random(d)) if diag else 0.9 * np.eye(d) + 0.3 * rng.standard_normal((d, d)) / np.sqrt(d)
        Q = np.diag(0.1 + rng.random(d)) if diag else 0.3 * np.eye(d) + 0.2 * rng.standard_normal((d, d)) / np.sqrt(d)
        Xp = rng.standard_normal((N, d))
        a = rng.integers(0, N, N).astype(np.uint32)
        step = int(rng.integers(1, 50))
        want = oracle.propagate(Xp, a, G, Q, dist, nu, 1.0, seed=seed, step=step)
        first = int(rng.integers(0, N))
        count = int(rng.integers(1, N - first + 1))
        Xd = torch.from_numpy(Xp).cuda()
        ad = torch.from_numpy(a[first:first + count].astype(np.int32)).cuda()
        out = torch.full((count + 1, d), float("nan"), dtype=torch.float64, device="cuda")
        cs.api.propagate_dev(Xd, ad, G, Q, out[:count], dist, nu, 1.0, seed=seed, step=step, first=first, ctx=ctx)
        torch.cuda.synchronize()
        tag = (seed, case, d, N, dist, diag, first, count)
        assert np.allclose(out[:count].cpu().numpy(), want[first:first + count], rtol=1e-9, atol=1e-9), tag
        assert bool(torch.isnan(out[count]).all()), tag


def test_box_muller_accuracy(cs, oracle):
    """The device Box-Muller uses its own ln / sincos(2 pi u) (kernels/smallops.h), the oracle libm's.
    With Q = I and mu = 0 a draw IS the normal variate: 2e5 of them agree to a few ulp of the largest
    value (|z| <= 8.6), far inside the 1e-9 the draw tests allow -- and the sample moments are right."""
    d, n = 2, 100_000
    D = cs.MultiVariateNormalDistribution(np.zeros(d), np.eye(d))
    got = D.sample(np.eye(d), 200, count=n, compat=False, seed=2024, step=9)
    want, _ = oracle.initialize(n, np.zeros(d), np.eye(d), "mvn", 0.0, 1.0, seed=2024, step=9)
    D.close()
    assert np.max(np.abs(got - want)) < 2e-14
    assert abs(got.mean()) < 0.01 and abs(got.var() - 1.0) < 0.01
    assert abs(np.mean(got[:, 0] * got[:, 1])) < 0.01


def test_R_level_draws(cs):
    x = cs.MVN([0.0, 0.0], np.eye(2))
    t = cs.MVT([0.0, 0.0, 0.0], np.eye(3), 3.0)
    assert x.shape == (2,) and t.shape == (3,) and np.all(np.isfinite(x)) and np.all(np.isfinite(t))
    assert not np.array_equal(cs.MVN([0.0, 0.0], np.eye(2)), x)   # successive calls differ


def test_unseeded_calls_are_independent_and_a_seed_reproduces_them(cs):
    """every unseeded R-level call draws a Philox key of its own (cusmc_stream_key(session
    seed, call counter)) -- two run() calls in a row are different replications, as with the
    reference's per-call std::random_device -- and set_seed(s) reproduces the whole sequence."""
    I = np.eye(2)
    Y = np.cumsum(0.1 * np.random.default_rng(1).standard_normal((2, 5)), axis=1)
    args = (300, 2, 5, Y, np.zeros(2), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn")

    def session():
        cs.set_seed(77)
        r1, r2 = cs.run(*args), cs.run(*args)
        a1 = cs.metropolis_hastings(np.arange(1.0, 101.0), 100, 10)
        a2 = cs.metropolis_hastings(np.arange(1.0, 101.0), 100, 10)
        l1 = cs.Sampler.metropolis_hastings_log(np.log(np.arange(1.0, 101.0)))
        l2 = cs.Sampler.metropolis_hastings_log(np.log(np.arange(1.0, 101.0)))
        t1, t2 = cs.MVT(np.zeros(3), np.eye(3), 3.0), cs.MVT(np.zeros(3), np.eye(3), 3.0)
        return r1, r2, a1, a2, l1, l2, t1, t2

    s1, s2 = session(), session()
    r1, r2, a1, a2, l1, l2, t1, t2 = s1
    assert not np.array_equal(r1["posterior_x"], r2["posterior_x"]) and not np.array_equal(r1["weights"], r2["weights"])
    assert not np.array_equal(a1, a2) and not np.array_equal(l1, l2) and not np.array_equal(t1, t2)
    for u, v in zip(s1, s2):   # the same seed gives the same sequence of calls
        if isinstance(u, dict):
            assert np.array_equal(u["posterior_x"], v["posterior_x"]) and np.array_equal(u["weights"], v["weights"])
        else:
            assert np.array_equal(u, v)
    cs.set_seed(2024)


def test_device_wrappers_reject_bad_tensors(cs):
    """shape / dtype / stride mistakes are ValueErrors, not out-of-bounds reads."""
    import torch
    from cusmc_amd import api
    D = cs.MultiVariateNormalDistribution(np.zeros(4), np.eye(4))
    X = torch.zeros(64, 6, dtype=torch.float64, device="cuda")
    out = torch.zeros(64, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        D.pdf_dev(X, out)                                    # wider than d: no silent first-d-columns
    with pytest.raises(ValueError):
        D.pdf_dev(X[:, :4].float(), out)                     # dtype
    with pytest.raises(ValueError):
        D.pdf_dev(X[:, :4], out, F=np.eye(3))                # F not d x d
    D.pdf_dev(X[:, :4], out)                                 # a strided view with ldx = 6 is fine
    Xp = torch.zeros(64, 4, dtype=torch.float64, device="cuda")
    a = torch.zeros(64, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        api.propagate_dev(Xp[:, :2], a, np.eye(2), np.eye(2), Xp[:, :2].contiguous())   # strided X_prev
    with pytest.raises(ValueError):
        api.propagate_dev(Xp, a[:10], np.eye(4), np.eye(4), Xp.clone())                 # len(a) != len(X_out)
    with pytest.raises(ValueError):
        api.propagate_dev(Xp, a, np.eye(3), np.eye(4), Xp.clone())                      # G not d x d
    with pytest.raises(ValueError):
        api.propagate_dev(Xp, a.long(), np.eye(4), np.eye(4), Xp.clone())               # int64 ancestors
    w = torch.ones(64, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        api.pf_step_dev(D, w[:10], Xp, np.eye(4), np.eye(4), np.zeros(4), None, a, Xp.clone(), w.clone())
    with pytest.raises(ValueError):
        cs.Sampler.metropolis_hastings_dev(w.float(), a)
    D.close()


def test_generic_fallback_raises_its_lds_limit_again(cs):
    """the staged generic kernel sizes its dynamic LDS from d; a first call above
    64 KB (d = 130) must not pin the limit for a later, larger d (250).  Reached through a row stride
    >= 2^24 doubles, which only the generic kernel serves."""
    import torch
    rng = np.random.default_rng(12)
    big = torch.zeros(3, 2 ** 24, dtype=torch.float64, device="cuda")
    out = torch.empty(3, dtype=torch.float64, device="cuda")
    for d in (130, 250):
        S = spd(rng, d)
        Xh = rng.standard_normal((3, d))
        big[:, :d] = torch.from_numpy(Xh).cuda()
        D = cs.MultiVariateNormalDistribution(np.zeros(d), S)
        D.pdf_dev(big[:, :d], out)
        want = D.pdf_batch(Xh)
        assert rel_err(out.cpu().numpy(), want) < 1e-9
        D.close()
    del big


def test_filter_against_golden(cs, golden):
    """d = 2, T = 10, N = 64 on the reference's example observations (data_raw/y_t.csv rows)."""
    I = np.eye(2)
    Y = golden["pf_y"].T  # run() takes d x T, columns = time (src/run.rcpp.cpp:91)
    for dist, nu in (("mvn", 0.0), ("mvt", 5.0)):
        out = cs.run(64, 2, Y.shape[1], Y, np.zeros(2), I, I, I, 0.5 * I, 0.1 * I, nu, "metropolis",
                     dist, B=10, seed=99, return_ancestors=True)
        assert out["weights"].shape == (10, 64) and out["posterior_x"].shape == (10, 64, 2)
        assert np.array_equal(out["ancestors"], golden["pf_%s_a" % dist])   # bit-exact indices
        assert np.allclose(out["posterior_x"], golden["pf_%s_X" % dist], rtol=1e-9, atol=1e-9)
        assert np.allclose(out["weights"], golden["pf_%s_w" % dist], rtol=1e-6)


@pytest.mark.parametrize("d,dist,nu", [(8, "mvn", 0.0), (3, "mvt", 4.0), (16, "mvn", 0.0), (24, "mvt", 3.0),
                                       (40, "mvt", 3.5), (70, "mvn", 0.0), (130, "mvn", 0.0)])
def test_filter_against_oracle_larger(cs, oracle, d, dist, nu):
    """run() against the oracle's MCMC() loop with dense F, G, V, W, C0 on every kernel family the
    time loop can take: the fused step (d <= 8), the three-launch step with the matrix-core proposal and
    the QL-rotated reweight (16 <= d <= 128, on and off the 16-grid), and the row-wise proposal with
    the wide log-pdf kernel (d > 128)."""
    rng = np.random.default_rng(4 + d)
    T, N = (6, 2000) if d <= 24 else (4, 600)
    Y = rng.standard_normal((T, d))
    G = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d / 8)
    F = np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d / 8)
    V, W, C0 = spd(rng, d), 0.3 * spd(rng, d), spd(rng, d)
    m0 = rng.standard_normal(d)
    out = cs.run(N, d, T, Y.T, m0, C0, F, G, V, W, nu, "metropolis", dist, seed=7, return_ancestors=True)
    X, w, a = oracle.pf_run(Y, N, m0, C0, F, G, V, W, dist, nu, B=10, seed=7, hoisted=True)
    # eigen square roots are unique only up to column order/sign; both sides run the same
    # Householder tridiagonalisation + implicit QL with the same ordering and sign convention, so
    # the factors coincide and trajectories can be compared directly
    # accept/reject sequences bit-exact (north_star): the weights agree to ~1e-13, so a flipped test
    # would need u within 1e-13 of a ratio -- any mismatch at all means something larger differs
    assert np.array_equal(out["ancestors"], a), "%d of %d ancestors differ" % (int(np.sum(out["ancestors"] != a)), a.size)
    assert np.allclose(out["posterior_x"], X, rtol=1e-8, atol=1e-8)
    assert np.allclose(out["weights"], w, rtol=1e-6, atol=1e-300)


@pytest.mark.parametrize("d,dist,nu", [(2, "mvn", 0.0), (7, "mvt", 4.0), (20, "mvn", 0.0), (64, "mvn", 0.0), (100, "mvt", 3.0),
                                       (112, "mvn", 0.0), (150, "mvn", 0.0)])
def test_filter_with_cholesky_factors(cs, oracle, monkeypatch, d, dist, nu):
    """CUSMC_PROPOSAL_FACTOR=cholesky: run() takes the lower Cholesky factors of C0 and W as the proposals' square
    roots instead of eigenSolver's V sqrt(Lambda) (src/mcmc.cpp:70-71, 280) -- the same law, and from d = 32 up the
    triangular proposal kernels.  Against the oracle's step functions composed into MCMC()'s loop with those
    factors; the default (eigen) form is what every other filter test covers.  An unknown value is refused."""
    rng = np.random.default_rng(40 + d)
    T, N = (6, 2000) if d <= 24 else (4, 600)
    Y = rng.standard_normal((T, d))
    G = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(max(d, 8) / 8)
    F = np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(max(d, 8) / 8)
    V, W, C0 = spd(rng, d), 0.3 * spd(rng, d), spd(rng, d)
    m0 = rng.standard_normal(d)
    monkeypatch.setenv("CUSMC_PROPOSAL_FACTOR", "cholesky")
    out = cs.run(N, d, T, Y.T, m0, C0, F, G, V, W, nu, "metropolis", dist, seed=9, return_ancestors=True)
    Q0, Qw = np.linalg.cholesky(C0), np.linalg.cholesky(W)
    X = np.zeros((T, N, d)); w = np.zeros((T, N)); a = np.zeros((T, N), np.uint32)
    X[0], w[0] = oracle.initialize(N, m0, Q0, dist, nu, 1.0, seed=9, step=0)
    for t in range(1, T):
        a[t] = oracle.metropolis(w[t - 1], 10, 9, step=t)
        X[t] = oracle.propagate(X[t - 1], a[t], G, Qw, dist, nu, 1.0, seed=9, step=t)
        w[t] = np.exp(oracle.logpdf_hoisted(Y[t] - X[t] @ F.T, np.zeros(d), V, None, dist, nu))
    assert np.array_equal(out["ancestors"], a)
    assert np.allclose(out["posterior_x"], X, rtol=1e-8, atol=1e-8)
    assert np.allclose(out["weights"], w, rtol=1e-6, atol=1e-300)
    # the sharded loop takes the same factors (two shards on device 0): bitwise the one-device run
    two = cs.run(N, d, T, Y.T, m0, C0, F, G, V, W, nu, "metropolis", dist, seed=9, return_ancestors=True, devices=[0, 0])
    for key in ("posterior_x", "weights", "ancestors"):
        assert np.array_equal(two[key], out[key]), key
    monkeypatch.setenv("CUSMC_PROPOSAL_FACTOR", "eigen")
    ref = cs.run(N, d, T, Y.T, m0, C0, F, G, V, W, nu, "metropolis", dist, seed=9, return_ancestors=True)
    assert not np.allclose(ref["posterior_x"][0], out["posterior_x"][0])  # (another square root: other realisations)
    monkeypatch.setenv("CUSMC_PROPOSAL_FACTOR", "qr")
    with pytest.raises(cs.CusmcError):
        cs.run(N, d, T, Y.T, m0, C0, F, G, V, W, nu, "metropolis", dist, seed=9)


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 4))
def test_filter_random_models(cs, oracle, seed):
    """Dispatch fuzz for run(): random d in [1, 140], Normal or Student-t, F = I or general, G and W
    diagonal or dense -- whatever kernels the time loop picks (fused step, observation table rows as
    shift or as rotated bias, diagonal / one-product / two-product proposals, row-wise above d = 128), the
    trajectories are the oracle's MCMC() loop's."""
    rng = np.random.default_rng(5000 + seed)
    for case in range(3):
        d = int(rng.choice([rng.integers(1, 9), rng.integers(9, 65), rng.integers(65, 141)]))
        dist = str(rng.choice(["mvn", "mvt"]))
        nu = float(rng.choice([3.0, 4.0, 7.5, 5.0])) if dist == "mvt" else 0.0
        T = int(rng.integers(2, 6))
        N = int(rng.integers(200, 1200)) if d <= 64 else int(rng.integers(100, 400))
        general_F, dense_G, dense_W = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        F = np.eye(d) + (0.05 * rng.standard_normal((d, d)) / np.sqrt(max(d, 8) / 8) if general_F else 0.0)
        G = np.diag(0.7 + 0.3 * rng.random(d)) + (0.05 * rng.standard_normal((d, d)) / np.sqrt(max(d, 8) / 8) if dense_G else 0.0)
        W = 0.3 * spd(rng, d) if dense_W else np.diag(0.1 + 0.3 * rng.random(d))
        V, C0 = spd(rng, d), spd(rng, d)
        m0 = rng.standard_normal(d)
        Y = rng.standard_normal((T, d))
        tag = (seed, case, d, dist, nu, T, N, general_F, dense_G, dense_W)
        out = cs.run(N, d, T, Y.T, m0, C0, F, G, V, W, nu, "metropolis", dist, seed=11 + case, return_ancestors=True)
        X, w, a = oracle.pf_run(Y, N, m0, C0, F, G, V, W, dist, nu, B=10, seed=11 + case, hoisted=True)
        assert np.array_equal(out["ancestors"], a), (tag, int(np.sum(out["ancestors"] != a)))
        assert np.allclose(out["posterior_x"], X, rtol=1e-8, atol=1e-8), tag
        assert np.allclose(out["weights"], w, rtol=1e-6, atol=1e-300), tag


@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 16])
@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 4.0)])
@pytest.mark.parametrize("general_F,diag_model", [(False, False), (True, False), (False, True)])
def test_fused_step_equals_three_launches(cs, d, dist, nu, general_F, diag_model):
    """cusmc_pf_step_dev (one launch for d <= 8) against resample -> propagate -> reweight through
    the separate entry points: ancestors identical, states and weights bitwise identical; also on
    a shard [first, first + count) of the chains."""
    import torch
    from cusmc_amd import api
    N, B, seed, step = 20_011, 10, 77, 3
    rng = np.random.default_rng(d + 10 * general_F)
    g = torch.Generator(device="cuda").manual_seed(d)
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    wp = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 1e-12
    G = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d))
    Q = 0.3 * np.eye(d) + 0.02 * rng.standard_normal((d, d))
    if diag_model:  # the separate propagate then takes the diagonal kernel: same values all the same
        G, Q = np.diag(np.diag(G)), np.diag(np.diag(Q))
    F = np.eye(d) + (0.1 * rng.standard_normal((d, d)) if general_F else 0.0)
    y = rng.standard_normal(d)
    V = spd(rng, d)
    obs = (cs.MultiVariateNormalDistribution(None, V) if dist == "mvn"
           else cs.MultiVariateTStudentDistribution(None, V, nu))
    obs.ctx.use_torch_stream()
    for first, count in ((0, N), (5_003, 9_999)):
        a1 = torch.empty(count, dtype=torch.int32, device="cuda")
        X1 = torch.empty(count, d, dtype=torch.float64, device="cuda")
        w1 = torch.empty(count, dtype=torch.float64, device="cuda")
        api.pf_step_dev(obs, wp, Xp, G, Q, y, F, a1, X1, w1, kind=dist, nu=nu, B=B, seed=seed, step=step, first=first)
        a2 = torch.empty_like(a1); X2 = torch.empty_like(X1); w2 = torch.empty_like(w1)
        cs.Sampler.metropolis_hastings_dev(wp, a2, B=B, t=step, seed=seed, first=first, ctx=obs.ctx)
        api.propagate_dev(Xp, a2, G, Q, X2, dist, nu, 1.0, seed=seed, step=step, first=first, ctx=obs.ctx)
        obs.reweight_dev(X2, y, F, w2, log=False)
        torch.cuda.synchronize()
        assert torch.equal(a1, a2)
        assert torch.equal(X1, X2)
        assert torch.equal(w1, w2)
    obs.close()


def test_fused_step_large_N(cs, oracle):
    """The fused step at N = 6e5 (resampling chain on the truncated weight table): ancestors equal the
    oracle's plain chain over the same weights, states and weights equal the three separate launches."""
    import torch
    from cusmc_amd import api
    N, d, B, seed, step = 600_000, 2, 10, 5, 4
    g = torch.Generator(device="cuda").manual_seed(3)
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    wp = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) ** 8 * 1e-30
    wp[::11] = 0.0
    I = np.eye(d)
    obs = cs.MultiVariateNormalDistribution(None, 0.5 * I)
    obs.ctx.use_torch_stream()
    a1 = torch.empty(N, dtype=torch.int32, device="cuda")
    X1 = torch.empty(N, d, dtype=torch.float64, device="cuda")
    w1 = torch.empty(N, dtype=torch.float64, device="cuda")
    y = np.array([0.1, -0.2])
    api.pf_step_dev(obs, wp, Xp, 0.9 * I, 0.3 * I, y, I, a1, X1, w1, B=B, seed=seed, step=step)
    a2 = torch.empty_like(a1); X2 = torch.empty_like(X1); w2 = torch.empty_like(w1)
    cs.Sampler.metropolis_hastings_dev(wp, a2, B=B, t=step, seed=seed, ctx=obs.ctx)
    api.propagate_dev(Xp, a2, 0.9 * I, 0.3 * I, X2, "mvn", 0.0, 1.0, seed=seed, step=step, ctx=obs.ctx)
    obs.reweight_dev(X2, y, I, w2, log=False)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(X1, X2) and torch.equal(w1, w2)
    want = oracle.metropolis(wp.cpu().numpy(), B, seed, step=step)
    assert np.array_equal(a1.cpu().numpy().astype(np.uint32), want)
    obs.close()


def test_filter_rejects_unknown_options(cs):
    I = np.eye(2)
    Y = np.zeros((2, 3))
    for res, dist in (("multinomial", "mvn"), ("metropolis", "normal")):
        with pytest.raises(cs.CusmcError) as e:
            cs.run(8, 2, 3, Y, np.zeros(2), I, I, I, I, I, 0.0, res, dist)
        assert e.value.code == 1 and "unknown" in str(e.value)


def test_filter_tracks_a_linear_gaussian_state(cs):
    """Statistical sanity at a real size: the filter mean follows the Kalman mean."""
    rng = np.random.default_rng(0)
    d, T, N = 2, 30, 200_000
    I = np.eye(d)
    V, W = 0.5 * I, 0.1 * I
    x = np.zeros(d)
    Y = np.zeros((T, d))
    for t in range(1, T):
        x = x + rng.multivariate_normal(np.zeros(d), W)
        Y[t] = x + rng.multivariate_normal(np.zeros(d), V)
    out = cs.run(N, d, T, Y.T, np.zeros(d), I, I, I, V, W, 0.0, "metropolis", "mvn", B=30, seed=3)
    m, P = np.zeros(d), I.copy()
    for t in range(1, T):
        P = P + W
        K = P @ np.linalg.inv(P + V)
        m = m + K @ (Y[t] - m)
        P = (I - K) @ P
    # posterior mean at T-1 = weighted mean of x_{T-1}
    w = out["weights"][-1]
    est = (out["posterior_x"][-1] * w[:, None]).sum(0) / w.sum()
    assert np.allclose(est, m, atol=0.05)


def test_cpp_host_mirror(cs):
    """cusmc_amd/host/cusmc_host.hpp -- the C++ mirror of the reference's classes -- through its
    own driver (tests/cpp/host_mirror_test.cpp, built by __graft_entry__.build())."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def _sharded_gpu_worker(rank, world, port, N, d, T, tmp, exchange=False):
    import os
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)  # rehearsal: both ranks on the one GPU of the test box
    from cusmc_amd.sharding import gather_final, gpu_filter_callables, gpu_filter_callables_exchange, run_filter_sharded
    I = np.eye(d)
    Y = np.cumsum(0.1 * np.random.default_rng(5).standard_normal((d, T)), axis=1)
    model = (Y, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, 0.0, "mvn")
    if exchange:
        init_fn, resample_fn, move_fn, obs = gpu_filter_callables_exchange(*model, B=10, seed=21)
        stats = {}
        Xl, wl, al = run_filter_sharded(N, T, init_fn, resample_fn=resample_fn, move_fn=move_fn, stats=stats)
        assert stats["row_bytes_in"] <= Xl.shape[1] * d * 8 * (T - 1)  # never more than its own N/R rows per step
    else:
        init_fn, step_fn, obs = gpu_filter_callables(*model, B=10, seed=21)
        Xl, wl, al = run_filter_sharded(N, T, init_fn, step_fn)
    torch.cuda.synchronize()
    Xf = gather_final(Xl.permute(1, 0, 2).contiguous().cpu(), N).permute(1, 0, 2)
    wf = gather_final(wl.t().contiguous().cpu(), N).t()
    af = gather_final(al.t().contiguous().cpu(), N).t()
    if rank == 0:
        np.savez(tmp, X=Xf.numpy(), w=wf.numpy(), a=af.numpy(), Y=Y)
    obs.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,exchange", [(5003, 2, False), (3001, 16, False), (5003, 2, True), (3001, 16, True),
                                          (2003, 64, True)])
def test_sharded_filter_on_gpu_equals_run(cs, tmp_path, N, d, exchange):
    """The multi-GPU filter path end to end (two ranks rehearsed on one device, gloo): shards
    computed with global Philox indices -- the concatenated history is bitwise the single-process
    cusmc_pf_run_host result.  exchange = False: weights and states all-gathered every step, one
    cusmc_pf_step_dev per rank; exchange = True: weights all-gathered, only the ancestor rows fetched
    from their owners (all-to-all), resample / propagate / reweight through their own entry points."""
    import socket
    import torch.multiprocessing as mp
    T = 5
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmp = str(tmp_path / "pf.npz")
    mp.spawn(_sharded_gpu_worker, args=(2, port, N, d, T, tmp, exchange), nprocs=2, join=True)
    got = np.load(tmp)
    I = np.eye(d)
    res = cs.run(N, d, T, got["Y"], np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn",
                 seed=21, return_ancestors=True)
    assert np.array_equal(got["a"][1:], res["ancestors"][1:].astype(got["a"].dtype))
    assert np.array_equal(got["X"], res["posterior_x"])
    assert np.array_equal(got["w"], res["weights"])


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,T,dist,nu,ndev", [(5003, 2, 6, "mvn", 0.0, 2), (3001, 16, 5, "mvn", 0.0, 2),
                                                (2003, 64, 4, "mvt", 4.0, 3), (40000, 8, 4, "mvn", 0.0, 2),
                                                (1201, 130, 3, "mvn", 0.0, 2)])
def test_multi_device_run_below_the_abi_equals_one_device(cs, N, d, T, dist, nu, ndev):
    """cusmc_pf_run_multi_host: the filter's time loop sharded over a device list BELOW the C ABI (one host
    thread + context per shard, weights exchanged by peer copies, only the ancestor rows fetched from their
    owners).  Rehearsed here with every shard on device 0; the history is bitwise the one-device
    cusmc_pf_run_host result (ragged shards, fused and unfused single-device paths, dense F and G)."""
    rng = np.random.default_rng(N + d)
    Y = np.cumsum(0.1 * rng.standard_normal((d, T)), axis=1)
    G = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d)
    F = np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d)
    V, W, C0 = spd(rng, d), 0.3 * spd(rng, d), spd(rng, d)
    m0 = rng.standard_normal(d)
    args = (N, d, T, Y, m0, C0, F, G, V, W, nu, "metropolis", dist)
    one = cs.run(*args, seed=31, return_ancestors=True)
    many = cs.run(*args, seed=31, return_ancestors=True, devices=[0] * ndev)
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], many[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [1, 300_000, 5_000_000])
def test_run_chunked_copy_out_equals_single_chunk(cs, monkeypatch, chunk):
    """cusmc_pf_run_host streams a history above 128 MB back in chunks of whole time steps on a second
    stream (per-chunk events, pre-faulting workers).  CUSMC_PF_CHUNK_BYTES forces that path on a small filter:
    one step per chunk, a few steps per chunk, and a ragged last chunk must all return the single-chunk history.
    (The 2.8 GB history of test_config2_filter_1e6_particles_T100 takes the same path at its natural size.)"""
    N, d, T = 6007, 4, 13
    rng = np.random.default_rng(9)
    Y = np.cumsum(0.1 * rng.standard_normal((d, T)), axis=1)
    I = np.eye(d)
    args = (N, d, T, Y, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn")
    one = cs.run(*args, seed=3, return_ancestors=True)
    monkeypatch.setenv("CUSMC_PF_CHUNK_BYTES", str(chunk))
    many = cs.run(*args, seed=3, return_ancestors=True)
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], many[k]), k
    partial = cs.run(*args, seed=3)  # (no ancestors requested: a NULL output pointer on the chunked path)
    assert np.array_equal(partial["posterior_x"], one["posterior_x"]) and np.array_equal(partial["weights"], one["weights"])


@pytest.mark.gpu
def test_cusmc_devices_environment_shards_run(cs, monkeypatch):
    """CUSMC_DEVICES in the environment routes the plain entry point (what rcpp/src/run.rcpp.cpp calls)
    through the multi-device loop; malformed lists are refused."""
    I = np.eye(2)
    Y = np.cumsum(0.1 * np.random.default_rng(2).standard_normal((2, 5)), axis=1)
    args = (4001, 2, 5, Y, np.zeros(2), I, I, 0.9 * I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn")
    one = cs.run(*args, seed=8, return_ancestors=True)
    monkeypatch.setenv("CUSMC_DEVICES", "0,0,0")
    env = cs.run(*args, seed=8, return_ancestors=True)
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], env[k]), k
    monkeypatch.setenv("CUSMC_DEVICES", "0;1")
    with pytest.raises(cs.CusmcError):
        cs.run(*args, seed=8)
    monkeypatch.setenv("CUSMC_DEVICES", "0,99")
    with pytest.raises(cs.CusmcError):
        cs.run(*args, seed=8)


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [0, 200_000])
def test_multi_device_run_chunked_copy_out(cs, monkeypatch, chunk):
    """The sharded loop's own copy-out: per-step contiguous copies of each shard's columns on a second stream,
    chunk by chunk behind the chunk's last step, one chunk and several ragged ones."""
    N, d, T = 6007, 4, 13
    rng = np.random.default_rng(9)
    Y = np.cumsum(0.1 * rng.standard_normal((d, T)), axis=1)
    I = np.eye(d)
    args = (N, d, T, Y, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn")
    one = cs.run(*args, seed=3, return_ancestors=True)
    if chunk:
        monkeypatch.setenv("CUSMC_PF_CHUNK_BYTES", str(chunk))
    many = cs.run(*args, seed=3, return_ancestors=True, devices=[0, 0, 0])
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], many[k]), k
    partial = cs.run(*args, seed=3, devices=[0, 0])
    assert np.array_equal(partial["posterior_x"], one["posterior_x"]) and np.array_equal(partial["weights"], one["weights"])
    single = cs.run(*args, seed=3, return_ancestors=True, devices=[0])   # a one-entry list: that device's library-owned context
    for k in ("ancestors", "posterior_x", "weights"):
        assert np.array_equal(one[k], single[k]), k


@pytest.mark.gpu
def test_multi_device_run_on_two_physical_gpus(cs):
    """the branches of cusmc_pf_run_multi_host that differ between DISTINCT devices (peer access, peer
    copies of the weights, gather_rows_sharded_kernel reading a peer's HBM, per-thread device binding) run only
    where two GPUs are visible -- the builder's boxes have one, the driver's node has eight."""
    from cusmc_amd import _lib
    if _lib.lib().cusmc_device_count() < 2:
        pytest.skip("one GPU visible: the cross-device branches stay unverified on hardware here")
    for N, d, T, dist, nu in ((50_003, 2, 8, "mvn", 0.0), (20_001, 64, 4, "mvt", 4.0)):
        rng = np.random.default_rng(N)
        Y = np.cumsum(0.1 * rng.standard_normal((d, T)), axis=1)
        G = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d)
        V, W, C0 = spd(rng, d), 0.3 * spd(rng, d), spd(rng, d)
        args = (N, d, T, Y, np.zeros(d), C0, np.eye(d), G, V, W, nu, "metropolis", dist)
        one = cs.run(*args, seed=5, return_ancestors=True)
        two = cs.run(*args, seed=5, return_ancestors=True, devices=[0, 1])
        for k in ("ancestors", "posterior_x", "weights"):
            assert np.array_equal(one[k], two[k]), (N, d, k)
    w = np.random.default_rng(1).random(200_000)
    assert np.array_equal(cs.Sampler.metropolis_hastings(w, B=50, seed=3), cs.Sampler.metropolis_hastings(w, B=50, seed=3, devices=[0, 1]))


@pytest.mark.gpu
@pytest.mark.parametrize("N,B,ndev", [(1_000_000, 100, 3), (300_000, 100, 3), (5003, 40, 4), (7, 1000, 3), (100_000, 1, 2)])
def test_multi_device_resampler_equals_one_device(cs, oracle, N, B, ndev):
    """(300 000 chains over three shards: each shard's 1e5 chains run the LDS-table kernel with a non-zero first chain.)
    cusmc_metropolis_multi_host (BASELINE configs[3] from the R boundary): the chains sharded
    over a device list below the C ABI, one host thread and library-owned context per shard, every device given
    the whole weight vector; ancestors bitwise those of one device -- and of the oracle --, for the density and
    the log-weight chain, ragged and tiny shards included.  Rehearsed with every shard on device 0."""
    rng = np.random.default_rng(N + B)
    w = np.exp(-0.5 * rng.chisquare(8, N)) * 1e-12
    one = cs.Sampler.metropolis_hastings(w, B=B, seed=77, t=3)
    many = cs.Sampler.metropolis_hastings(w, B=B, seed=77, t=3, devices=[0] * ndev)
    assert np.array_equal(one, many)
    if N <= 300_000:
        assert np.array_equal(one, oracle.metropolis(w, B, 77, step=3))
    lw = np.log(w)
    assert np.array_equal(cs.Sampler.metropolis_hastings_log(lw, B=B, seed=78, t=2),
                          cs.Sampler.metropolis_hastings_log(lw, B=B, seed=78, t=2, devices=[0] * ndev))


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,dist,nu,ndev", [(100_000, 256, "mvn", 0.0, 3), (5003, 64, "mvt", 4.0, 4), (4099, 3, "mvn", 0.0, 3),
                                              (20_011, 65, "mvn", 0.0, 2), (300, 8, "mvn", 0.0, 3)])
def test_multi_device_density_equals_one_device(cs, N, d, dist, nu, ndev):
    """cusmc_dist_pdf_multi_host / cusmc_dist_reweight_multi_host: the ROWS of a host batch sharded over a device
    list, a replica of the distribution per shard; every row bitwise the one-device value (pdf(y, F) with mu != 0
    and a dense F, reweight_G with a dense F, densities and log-densities), ragged shards, a batch too small to
    shard, and the replicas reused by a second call."""
    rng = np.random.default_rng(N + d)
    sigma, mu = spd(rng, d), rng.standard_normal(d)
    X = mu + rng.standard_normal((N, d))
    F = np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    y = rng.standard_normal(d)
    D = (cs.MultiVariateNormalDistribution(mu, sigma) if dist == "mvn" else cs.MultiVariateTStudentDistribution(mu, sigma, nu))
    devs = [0] * ndev
    for log in (True, False):
        assert np.array_equal(D.pdf_batch(X, None, log=log), D.pdf_batch(X, None, log=log, devices=devs))
        assert np.array_equal(D.pdf_batch(X, F, log=log), D.pdf_batch(X, F, log=log, devices=devs))
        assert np.array_equal(D.reweight(X, y, F, log=log), D.reweight(X, y, F, log=log, devices=devs))
    assert np.array_equal(D.reweight(X, y, None), D.reweight(X, y, None, devices=devs[:2]))
    D.close()


@pytest.mark.gpu
def test_cusmc_devices_environment_shards_the_host_entry_points(cs, monkeypatch):
    """CUSMC_DEVICES routes cusmc_metropolis_host, cusmc_metropolis_log_host, cusmc_dist_pdf_host and
    cusmc_dist_reweight_host -- what rcpp/src/samplers.rcpp.cpp and rcpp/src/mv*_dist.rcpp.cpp bind -- over the
    list; a one-entry list naming the context's own device is the plain path; the caller's current device is
    left as it was."""
    import torch
    rng = np.random.default_rng(4)
    N, d = 30_011, 32
    sigma = spd(rng, d)
    X = rng.standard_normal((N, d))
    w = rng.random(N)
    D = cs.MultiVariateNormalDistribution(None, sigma)
    lp, a = D.pdf_batch(X), cs.Sampler.metropolis_hastings(w, B=20, seed=5)
    al = cs.Sampler.metropolis_hastings_log(np.log(w), B=20, seed=6)
    for env in ("0,0,0", "0"):
        monkeypatch.setenv("CUSMC_DEVICES", env)
        dev_before = torch.cuda.current_device()
        assert np.array_equal(lp, D.pdf_batch(X))
        assert np.array_equal(a, cs.Sampler.metropolis_hastings(w, B=20, seed=5))
        assert np.array_equal(al, cs.Sampler.metropolis_hastings_log(np.log(w), B=20, seed=6))
        assert np.array_equal(D.reweight(X, X[0], None), D.reweight(X, X[0], None, devices=[0, 0]))
        assert torch.cuda.current_device() == dev_before
    monkeypatch.setenv("CUSMC_DEVICES", "0,x")
    with pytest.raises(cs.CusmcError):
        D.pdf_batch(X)
    with pytest.raises(cs.CusmcError):
        cs.Sampler.metropolis_hastings(w, B=20, seed=5)
    monkeypatch.delenv("CUSMC_DEVICES")
    D.close()


# --- per-particle covariances ---------------------------------------------

def _random_covariances(rng, N, d):
    A = rng.standard_normal((N, d, d))
    scale = np.exp(rng.uniform(-2, 2, size=(N, 1, 1)))
    return scale * (A @ np.transpose(A, (0, 2, 1)) / d + np.eye(d))


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 11, 16])
def test_batched_cholesky_bit_exact(cs, oracle, d):
    """cusmc_chol_batched_host against the oracle's restatement of the same operation order:
    factors bit for bit (fma chains, one sqrt, one divide per element), log-determinants to the
    last bits of two different ln, the pivot report for matrices that are not positive definite."""
    rng = np.random.default_rng(300 + d)
    N = 1000 + d  # not a multiple of the 64-lane workgroup
    S = _random_covariances(rng, N, d)
    bad = [0, 17, N - 1]
    for n, i in enumerate(bad):
        S[i] = np.eye(d)
        S[i][min(n, d - 1), min(n, d - 1)] = -2.0 if n else 0.0
    L, logdet, info = cs.api.cholesky_batched(S)
    Lo, ldo, io = oracle.chol_batched(S)
    assert np.array_equal(info, io)
    assert info[bad[0]] == 1 and set(np.flatnonzero(info)) == set(bad)
    good = info == 0
    assert np.array_equal(L[good], Lo[good])
    assert np.all(np.triu(L[good], 1) == 0.0)
    assert np.allclose(logdet[good], ldo[good], rtol=1e-14, atol=1e-13)
    assert np.all(np.isnan(logdet[~good]))
    with pytest.raises(cs.CusmcError):
        cs.api.cholesky_batched(np.tile(np.eye(17), (3, 1, 1)))
    assert cs.api.cholesky_batched(np.empty((0, d, d)))[0].shape == (0, d, d)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 4, 8, 13, 16])
@pytest.mark.parametrize("dist", ["mvn", "mvt"])
def test_logpdf_per_particle_covariance(cs, oracle, d, dist):
    """cusmc_logpdf_percov_host against the reference's pdf() applied with a distribution object
    per particle (oracle.pdf_percov: LU determinant + inverse per particle), tolerance 1e-6 relative
    on the log-density as for the shared-covariance kernels; means per particle, shared, or absent."""
    rng = np.random.default_rng(400 + d)
    N = 777
    S = _random_covariances(rng, N, d)
    X = rng.standard_normal((N, d))
    nu = None if dist == "mvn" else 4.0
    for mu in (rng.standard_normal((N, d)), rng.standard_normal(d), None):
        lp, info = cs.api.logpdf_percov(X, mu, S, nu=nu)
        assert not info.any()
        want = np.log(oracle.pdf_percov(X, mu, S, dist=dist, nu=nu or 0.0))
        assert rel_err(lp, want) < RTOL
        assert np.abs(lp - want).max() < 1e-9 * max(1.0, np.abs(want).max())
    dens, _ = cs.api.logpdf_percov(X, None, S, nu=nu, log=False)
    assert rel_err(dens, oracle.pdf_percov(X, None, S, dist=dist, nu=nu or 0.0)) < RTOL
    S[5] = -np.eye(d)
    lp, info = cs.api.logpdf_percov(X, None, S, nu=nu)
    assert info[5] == 1 and np.isnan(lp[5]) and np.isfinite(np.delete(lp, 5)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("d", [16, 64, 100, 200])
def test_far_centre_keeps_exact_subtraction(cs, oracle, d):
    """The centre is subtracted from the particle BEFORE the factor is applied, as the reference's
    r = y - F mu (src/statistics.cc.cpp:192): a distribution centred 1e12 standard deviations from
    the origin is evaluated as accurately as one at the origin.  (Folding the centre into a bias,
    z = W x - W c, would be free on the matrix cores -- the subtraction costs 3 - 6 % of a launch at
    d = 64 -- but is wrong in the third digit here and gives up the bitwise shift identity
    test_full_size_properties checks; measured and not adopted)"""
    rng = np.random.default_rng(d)
    sigma = spd(rng, d)
    N = 500
    for centre in (1e12, 30.0):
        mu = centre * (1.0 + 0.1 * rng.random(d))
        Xh = mu + rng.standard_normal((N, d))
        D = cs.MultiVariateNormalDistribution(mu, sigma)
        assert rel_err(D.pdf_batch(Xh), oracle.logpdf_hoisted(Xh, mu, sigma, None, "mvn", 0.0)) < RTOL
        # reweight_G with F = I is the same form about y
        Z = cs.MultiVariateNormalDistribution(None, sigma)
        assert rel_err(Z.reweight(Xh, mu, np.eye(d)), oracle.logpdf_hoisted(mu[None, :] - Xh, None, sigma, None, "mvn", 0.0)) < RTOL
        D.close()
        Z.close()


@pytest.mark.gpu
def test_interpreter_exit_with_live_handles(cs):
    """A process that ends with distribution handles still alive -- normally, or through an uncaught
    exception -- exits cleanly: the context closes its distributions first whatever order the module
    globals are torn down in (a handle outliving its context used to abort the interpreter at exit)."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "scripts", "exit_check.py")
    ok = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    exc = subprocess.run([sys.executable, script, "x"], capture_output=True, text=True, timeout=300)
    assert exc.returncode == 1 and "RuntimeError" in exc.stderr and "terminate called" not in exc.stderr, exc.stderr[-2000:]


@pytest.mark.gpu
def test_context_destroyed_before_its_distribution(cs):
    """C-ABI lifetime rule: destroying a context orphans the distributions created on it -- their
    device buffers go with it, a later call through such a handle fails with a status instead of
    touching freed memory, and the handle can still be destroyed (finalizers run in any order)."""
    ctx = cs.Context()
    D = cs.MultiVariateNormalDistribution(np.zeros(20), np.eye(20), ctx=ctx)
    assert np.isfinite(D.pdf_batch(np.zeros((3, 20)))).all()
    ctx._distributions.discard(D)  # bypass the Python-level ordering: exercise the library's own
    ctx.close()
    with pytest.raises(cs.CusmcError, match="destroyed"):
        D.pdf_batch(np.zeros((3, 20)))
    D.close()
    E = cs.MultiVariateNormalDistribution(np.zeros(20), np.eye(20))  # the default context is unaffected
    assert np.isfinite(E.pdf_batch(np.zeros((3, 20)))).all()
    E.close()
