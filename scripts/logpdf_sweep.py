#!/usr/bin/env python3
"""Times the log-pdf kernels over d and distribution kind on one MI355X (device-resident X, HIP
events on the launch stream).  Developer aid for A/B runs between library builds:
    python scripts/logpdf_sweep.py [tag]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402


def spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(ts))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    ctx = cusmc_amd.api.default_context().use_torch_stream()
    g = torch.Generator(device="cuda").manual_seed(7)
    cells = []
    for d in (16, 32, 48, 64, 80, 96, 112):
        N = 64_000_000 // d
        X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        out = torch.empty(N, dtype=torch.float64, device="cuda")
        mu = np.zeros(d)
        for kind in ("mvn", "mvn+mu", "mvt", "reweight"):
            if kind == "mvt":
                D = cusmc_amd.MultiVariateTStudentDistribution(mu, spd(d, 1), 4.0, ctx=ctx)
            else:
                D = cusmc_amd.MultiVariateNormalDistribution(mu + (0.1 if kind == "mvn+mu" else 0.0), spd(d, 1), ctx=ctx)
            if kind == "reweight":
                F = np.eye(d) + 0.01 * np.random.default_rng(2).standard_normal((d, d))
                y = np.zeros(d)
                fn = lambda: D.reweight_dev(X, y, F, out)  # noqa: E731
            else:
                fn = lambda: D.pdf_dev(X, out)  # noqa: E731
            t = timed(fn, 200, 300)
            cells.append("d=%d %s %.1f us (%.2f TB/s)" % (d, kind, t, N * (8 * d + 8) / t / 1e6))
            D.close()
        del X, out
    print(tag, " | ".join(cells))


if __name__ == "__main__":
    main()
