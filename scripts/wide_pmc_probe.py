#!/usr/bin/env python3
"""A few launches of the wide log-pdf kernel per block count (and of the d = 128 / 176 tile kernel beside it) for
scripts/wide_pmc.sh's rocprofv3 --pmc passes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402

ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
for d in (128, 176, 192, 208, 224, 240, 256):
    N = int(1.28e8 // d)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    A = rng.standard_normal((d, d))
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), A @ A.T / d + np.eye(d), ctx=ctx)
    for _ in range(6):
        D.pdf_dev(X, out)
    torch.cuda.synchronize()
    D.close()
    del X, out
