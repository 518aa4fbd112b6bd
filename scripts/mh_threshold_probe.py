#!/usr/bin/env python3
"""Metropolis resampler rate over N and B (device-resident weights).  Developer aid: where does the
truncated-table chain (CUSMC_MH_HI=1 forces it, =0 forbids it in calibration builds) pay?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402

ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for N, B in ((1000, 10), (10_000, 10), (100_000, 10), (100_000, 1000), (300_000, 10), (300_000, 100), (1_000_000, 10)):
    w = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 1e-20
    a = torch.empty(N, dtype=torch.int32, device="cuda")
    st = [0]

    def f():
        st[0] += 1
        cusmc_amd.Sampler.metropolis_hastings_dev(w, a, B=B, t=st[0], seed=1, first=0, ctx=ctx)
    reps = 5 if B >= 100 else 50
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print("%s N=%d B=%d: %.1f us, %.3g steps/s" % (os.environ.get("CUSMC_MH_HI", "default"), N, B, t * 1e6, N * B / t), flush=True)
