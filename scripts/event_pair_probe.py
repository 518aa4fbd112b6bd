#!/usr/bin/env python3
"""Three ways to time the headline launch with HIP events, K = 20 launches after 600 settle launches:
(a) one event pair around all K (what bench.py did in round 1: includes the K - 1 gaps between launches),
(b) an event pair around EACH launch (the kernel's own duration, what rocprofv3 --kernel-trace reports),
(c) wall clock around the region."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cusmc_amd
from bench import make_sigma

N, D, K = 1_000_000, 64, 20
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1234)
X = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
out = torch.empty(N, dtype=torch.float64, device="cuda")
mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), make_sigma(D, 1))
mvn.ctx.use_torch_stream()
f = lambda: mvn.pdf_dev(X, out)
for rnd in range(4):
    for _ in range(600):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        f()
    e1.record()
    torch.cuda.synchronize()
    wall_a = (time.perf_counter() - t0) / K * 1e6
    span = e0.elapsed_time(e1) / K * 1e3
    for _ in range(600):
        f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    wall_b = (time.perf_counter() - t0) / K * 1e6
    per = [a.elapsed_time(b) * 1e3 for a, b in ev]
    total = ev[0][0].elapsed_time(ev[-1][1]) / K * 1e3
    print("round %d: (a) one pair: %.2f us/launch (wall %.1f) | (b) pair per launch: mean %.2f min %.2f max %.2f us, "
          "first-to-last span %.2f us/launch (wall %.1f)" % (rnd, span, wall_a, sum(per) / K, min(per), max(per), total, wall_b), flush=True)
