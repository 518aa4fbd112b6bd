"""Proposal-draw timings with DENSE Q over d, G dense and G diagonal (16e6/d particles).  Developer aid."""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
for d in (2, 8, 16, 32, 48, 64, 65, 80, 100, 128):
    N = 16_000_000 // d
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    G = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    Q = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    st = [0]
    for kind, nu in (("mvn", 0.0), ("mvt", 4.0)):
        for label, Gm in (("dense G", G), ("diagonal G", np.diag(np.diag(G)))):
            def f():
                st[0] += 1
                cusmc_amd.api.propagate_dev(Xp, anc, Gm, Q, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
            t = timed(f, 5, 2)
            print("d=%d N=%d %s, dense Q, %s %.1f us (%.2f TB/s algorithmic)" % (d, N, label, kind, t, N * (16 * d + 4) / t / 1e6), flush=True)
    del Xp, anc, out
