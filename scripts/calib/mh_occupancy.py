"""The resampler at B = 1000 for chain counts around BASELINE configs[1]'s 1e5: 65536 chains are one wave per SIMD,
131072 two -- is the 1e5-chain launch (1.53 waves per SIMD) paying for two?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(3)
B = 1000
for N in (32768, 65536, 100_000, 131072, 196608, 262144):
    w = torch.exp(-0.5 * (torch.randn(N, 32, dtype=torch.float64, device="cuda", generator=g) ** 2).sum(1)) * 1e-13
    a = torch.empty(N, dtype=torch.int32, device="cuda")
    st = [0]
    def f():
        st[0] += 1
        cusmc_amd.Sampler.metropolis_hastings_dev(w, a, B=B, t=st[0], seed=1, ctx=ctx)
    t = timed(f, 5, 2)
    print("N=%d chains (%.2f waves per SIMD) B=%d: %.1f us = %.3g steps/s" % (N, N / 65536, B, t, N * B / t * 1e6), flush=True)
