#!/usr/bin/env python3
"""A/B of the RNG-bound kernels between the library and a calibration build (libcusmc_hip_exp.so):
    python scripts/rng_ab.py; EXP=1 python scripts/rng_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cusmc_amd import _lib  # noqa: E402
if os.environ.get("EXP"):
    _lib.SO_PATH = _lib.SO_PATH.replace("libcusmc_hip.so", "libcusmc_hip_exp.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusmc_amd  # noqa: E402
from scripts.logpdf_sweep import timed  # noqa: E402

ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
tag = "exp " if os.environ.get("EXP") else "base"
for d, dense in ((64, False), (64, True), (8, False), (2, False)):
    N = 1_000_000
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    G = 0.9 * np.eye(d)
    Q = 0.3 * np.eye(d) + (0.05 * np.random.default_rng(0).standard_normal((d, d)) if dense else 0.0)
    st = [0]

    def f():
        st[0] += 1
        cusmc_amd.api.propagate_dev(Xp, anc, G, Q, out, "mvn", 0.0, 1.0, seed=1, step=st[0], ctx=ctx)
    print("%s propagate d=%d %s Q: %.1f us" % (tag, d, "dense" if dense else "diagonal", timed(f, 20, 5)), flush=True)
    del Xp, anc, out
I = np.eye(2)
N = 1_000_000
Xp = torch.randn(N, 2, dtype=torch.float64, device="cuda", generator=g)
wp = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
a = torch.empty(N, dtype=torch.int32, device="cuda")
Xo = torch.empty(N, 2, dtype=torch.float64, device="cuda")
wo = torch.empty(N, dtype=torch.float64, device="cuda")
obs = cusmc_amd.MultiVariateNormalDistribution(None, 0.5 * I, ctx=ctx)
st = [0]


def fused():
    st[0] += 1
    cusmc_amd.api.pf_step_dev(obs, wp, Xp, I, 0.3 * I, np.zeros(2), I, a, Xo, wo, B=10, seed=1, step=st[0])


print("%s fused filter step N=1e6 d=2: %.1f us" % (tag, timed(fused, 50, 10)), flush=True)
