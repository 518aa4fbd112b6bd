"""NOT part of the suite (pytest.ini's testpaths = tests: a bare `pytest` never collects it).  Round 1 saw `terminate called after throwing an
instance of 'std::bad_variant_access'` when pytest exited after a FAILED gpu test with live handles and device
tensors in the failing frame (gpurun_out/pytest.txt).  This re-creates that exit on purpose:
    CUSMC_TRACE_TERMINATE=1 python -m pytest scripts/exit_repro_pytest -x -q -p no:cacheprovider
and the terminate tracer in libcusmc_hip prints the stack that threw, if it happens again."""
import numpy as np
import pytest


def spd(rng, d):
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


@pytest.fixture(scope="module")
def cs():
    import cusmc_amd
    cusmc_amd.set_seed(2024)
    return cusmc_amd


def test_passes_first_with_the_oracle_loaded(cs, oracle):
    w = np.random.default_rng(0).random(1000)
    assert np.array_equal(cs.Sampler.metropolis_hastings(w, 1000, t=1, B=10, seed=5), oracle.metropolis(w, 10, 5, step=1))
    x = cs.MVN([0.0, 0.0], np.eye(2))
    assert x.shape == (2,)


@pytest.mark.parametrize("dist,nu", [("mvn", 0.0), ("mvt", 4.0)])
def test_fails_with_live_handles(cs, dist, nu):
    import torch
    from cusmc_amd import api
    rng = np.random.default_rng(3)
    d, N, B, seed, step = 2, 20000, 10, 5, 3
    G, Q = 0.9 * np.eye(d), 0.3 * np.eye(d)
    y = rng.standard_normal(d)
    wp = torch.rand(N, dtype=torch.float64, device="cuda")
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda")
    obs = (cs.MultiVariateNormalDistribution(None, spd(rng, d)) if dist == "mvn"
           else cs.MultiVariateTStudentDistribution(None, spd(rng, d), nu))
    obs.ctx.use_torch_stream()
    a1 = torch.empty(N, dtype=torch.int32, device="cuda")
    X1 = torch.empty(N, d, dtype=torch.float64, device="cuda")
    w1 = torch.empty(N, dtype=torch.float64, device="cuda")
    api.pf_step_dev(obs, wp, Xp, G, Q, y, None, a1, X1, w1, kind=dist, nu=nu, B=B, seed=seed, step=step)
    torch.cuda.synchronize()
    assert torch.equal(X1, X1 + 1)   # fails on purpose, with obs / tensors alive in this frame
