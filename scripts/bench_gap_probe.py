#!/usr/bin/env python3
"""Why does bench.py's event-timed launch read ~3 us above scripts/calib/ablate's on the same box?
Times the headline launch from Python with the timed-region length, the input data and the call path varied."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cusmc_amd
from cusmc_amd import _lib
from bench import make_sigma

N, D = 1_000_000, 64
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1234)
Xn = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
Xu = (torch.rand(N, D, dtype=torch.float64, device="cuda", generator=g) - 0.5) * 4
out = torch.empty(N, dtype=torch.float64, device="cuda")
mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), make_sigma(D, 1))
near_I = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), np.eye(D) + 0.01 * make_sigma(D, 3))
mvn.ctx.use_torch_stream()
L = _lib.lib()


def timed(fn, K, settle=600):
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        fn()
    host = (time.perf_counter() - t0) / K * 1e6
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3, host


def raw(dist, X):
    h, xp, op = dist._h, C.c_void_p(X.data_ptr()), C.c_void_p(out.data_ptr())
    f = L.cusmc_dist_pdf_dev
    return lambda: f(h, xp, N, D, None, 0, op)


for rnd in range(2):
    for name, fn in (("api, randn X, bench Sigma", lambda: mvn.pdf_dev(Xn, out)),
                     ("raw ctypes, randn X, bench Sigma", raw(mvn, Xn)),
                     ("raw ctypes, uniform X, bench Sigma", raw(mvn, Xu)),
                     ("raw ctypes, randn X, Sigma near I", raw(near_I, Xn)),
                     ("raw ctypes, zeros X, bench Sigma", raw(mvn, torch.zeros_like(Xn)))):
        row = []
        for K in (20, 200, 2000):
            us, host = timed(fn, K)
            row.append("K=%d: %.1f us (host %.1f)" % (K, us, host))
        print("%-36s %s" % (name, " | ".join(row)), flush=True)
