"""Times the resampler at BASELINE configs[1] (1e5 chains x 1e3 iterations) for the library in place."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(3)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for N, B in ((100_000, 1000), (1_000_000, 100), (1_000_000, 10)):
    w = torch.exp(-0.5 * (torch.randn(N, 32, dtype=torch.float64, device="cuda", generator=g) ** 2).sum(1)) * 1e-13
    a = torch.empty(N, dtype=torch.int32, device="cuda")
    st = [0]
    def f():
        st[0] += 1
        cusmc_amd.Sampler.metropolis_hastings_dev(w, a, B=B, t=st[0], seed=1, ctx=ctx)
    t = timed(f, 5, 2)
    print("%s N=%d B=%d: %.1f us = %.3g steps/s" % (tag, N, B, t, N * B / t * 1e6), flush=True)
