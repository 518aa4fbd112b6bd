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


def cycle(iterations):
    for it in range(iterations):
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
    return torch.cuda.mem_get_info()[0]


# a first pass loads every code object and grows the context's scratch buffers to their working size (that is not a
# leak and it saturates); the passes after it must not take anything more
free = [torch.cuda.mem_get_info()[0]]
for _ in range(3):
    free.append(cycle(2000))
print("free device memory (MB): start %.1f | after pass 1 %.1f | pass 2 %.1f | pass 3 %.1f" % tuple(f / 1e6 for f in free))
print("taken by pass 1: %.1f MB (code objects, scratch); by passes 2 and 3: %.1f MB" % ((free[0] - free[1]) / 1e6, (free[1] - free[3]) / 1e6))
assert free[1] - free[3] < 16e6, "device memory is leaking"
