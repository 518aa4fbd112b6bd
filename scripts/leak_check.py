#!/usr/bin/env python3
"""Device-memory leak smoke: many create / use / destroy cycles through the R-level API with changing
parameters (so the handle caches evict), free device memory before and after."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402

rng = np.random.default_rng(0)
cusmc_amd.MVNPDF(np.zeros(4), np.zeros(4), np.eye(4))
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for it in range(3000):
    d = int(rng.integers(1, 80))
    A = rng.standard_normal((d, d))
    S = A @ A.T / d + np.eye(d)
    cusmc_amd.MVNPDF(rng.standard_normal(d), np.zeros(d), S)
    if it % 3 == 0:
        cusmc_amd.MVT(np.zeros(d), S, 4.0)
    if it % 50 == 0:
        Y = rng.standard_normal((d, 4))
        cusmc_amd.run(200, d, 4, Y, np.zeros(d), S, np.eye(d), np.eye(d), S, S, 0.0, "metropolis", "mvn", seed=it)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free device memory: before %.1f MB, after %.1f MB (delta %.1f MB)" % (free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6))
assert free0 - free1 < 64e6, "device memory is leaking"
