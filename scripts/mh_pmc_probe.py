#!/usr/bin/env python3
"""A few launches of the Metropolis resampler for a rocprofv3 --pmc pass: BASELINE configs[1] (N = 1e5 chains, B = 1e3,
weights = d = 32 MVN densities: metropolis_kernel, table in L2) and the filter's shape (N = 1e6, B = 10 and B = 100:
hiword_kernel + metropolis_hi_kernel, 4 MB table).   rocprofv3 --pmc <counters> ... -- python3 scripts/mh_pmc_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402
import bench  # noqa: E402

ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
sig = bench.make_sigma(32, 2)
d32 = cusmc_amd.MultiVariateNormalDistribution(np.zeros(32), sig, ctx=ctx)
Xw = (torch.randn(100_000, 32, dtype=torch.float64, device="cuda", generator=g) @ torch.from_numpy(np.linalg.cholesky(sig).T).cuda()).contiguous()
w = torch.empty(100_000, dtype=torch.float64, device="cuda")
d32.pdf_dev(Xw, w, log=False)
a = torch.empty(100_000, dtype=torch.int32, device="cuda")
for t in range(1, 4):
    cusmc_amd.Sampler.metropolis_hastings_dev(w, a, B=1000, t=t, seed=1, first=0, ctx=ctx)
torch.cuda.synchronize()
w6 = torch.rand(1_000_000, dtype=torch.float64, device="cuda", generator=g) * 1e-20
a6 = torch.empty(1_000_000, dtype=torch.int32, device="cuda")
for B in (10, 100):
    for t in range(1, 4):
        cusmc_amd.Sampler.metropolis_hastings_dev(w6, a6, B=B, t=t, seed=1, first=0, ctx=ctx)
torch.cuda.synchronize()
