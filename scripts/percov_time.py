#!/usr/bin/env python3
"""Times the per-particle-covariance kernels (batched Cholesky, per-covariance log-density),
device-resident, over d.  Developer aid."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402
from cusmc_amd import _lib  # noqa: E402


def main():
    ctx = cusmc_amd.api.default_context().use_torch_stream()
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    for d in (2, 4, 8, 16):
        N = 1_000_000
        A = torch.randn(N, d, d, dtype=torch.float64, device="cuda", generator=g)
        S = (A @ A.transpose(1, 2) / d + torch.eye(d, dtype=torch.float64, device="cuda")).contiguous()
        X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        Lo = torch.empty_like(S)
        ld = torch.empty(N, dtype=torch.float64, device="cuda")
        info = torch.empty(N, dtype=torch.int32, device="cuda")
        out = torch.empty(N, dtype=torch.float64, device="cuda")
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731

        def chol():
            _lib.check(L.cusmc_chol_batched_dev(ctx._h, p(S), N, d, p(Lo), p(ld), p(info)))

        def lpdf():
            _lib.check(L.cusmc_logpdf_percov_dev(ctx._h, _lib.MVN, 0.0, p(X), N, d, None, 0, p(S), d, _lib.OUT_LOG,
                                                 p(out), p(info)))
        for name, fn, bytes_ in (("cholesky", chol, N * (16 * d * d + 12)), ("logpdf", lpdf, N * (8 * d * d + 8 * d + 12))):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20 * 1e-3
            print("d=%2d N=%d %-8s %8.1f us  %6.2f TB/s algorithmic  %.3g matrices/s" % (d, N, name, t * 1e6, bytes_ / t / 1e12, N / t),
                  flush=True)


if __name__ == "__main__":
    main()
