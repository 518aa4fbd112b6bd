"""Proposal draws for 128 < d <= 256 (kernels/propagate_mfma_wide.hip) next to the reweight of the same batch:
VERDICT r01 item 6 asks for 5e5 x 256 dense within 2x of the reweight.  Developer aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cusmc_amd
from scripts.logpdf_sweep import timed

ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
for d, N in ((256, 500_000), (192, 333_333), (144, 444_444), (130, 400_000)):
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    w = torch.empty(N, dtype=torch.float64, device="cuda")
    A = rng.standard_normal((d, d))
    V = A @ A.T / d + np.eye(d)
    G = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    F = np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    Q = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    obs = cusmc_amd.MultiVariateNormalDistribution(None, V, ctx=ctx)
    y = rng.standard_normal(d)
    t_rw = timed(lambda: obs.reweight_dev(Xp, y, F, w), 5, 2)
    print("d=%d N=%d reweight (dense F): %.1f us" % (d, N, t_rw), flush=True)
    st = [0]
    for kind, nu in (("mvn", 0.0), ("mvt", 4.0)):
        for label, Gm in (("dense G", G), ("diagonal G", np.diag(np.diag(G)))):
            def f():
                st[0] += 1
                cusmc_amd.api.propagate_dev(Xp, anc, Gm, Q, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
            t = timed(f, 5, 2)
            nb = (d + 15) // 16
            flop = (2 if label == "dense G" else 1) * 2.0 * (16 * nb) ** 2
            print("d=%d N=%d propagate %s, dense Q, %s: %.1f us = %.2fx the reweight, %.1f TFLOP/s" %
                  (d, N, label, kind, t, t / t_rw, N * flop / t / 1e6), flush=True)
    obs.close()
    del Xp, anc, out, w
