#!/usr/bin/env python3
"""A few launches of every matrix-core kernel, for a rocprofv3 --pmc pass (MfmaUtil, LDS bank conflicts):
    rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d OUT -- python3 scripts/mfma_util_probe.py"""
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


ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for N, d in ((1_000_000, 64), (1_000_000, 32), (1_000_000, 128), (500_000, 192), (500_000, 208), (500_000, 224), (500_000, 240),
             (500_000, 256)):  # (the wide kernel at every block count it serves: NB = 12 .. 16)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
    T = cusmc_amd.MultiVariateTStudentDistribution(np.zeros(d), spd(d, 1), 4.0, ctx=ctx)
    F = np.eye(d) + 0.3 * np.random.default_rng(2).standard_normal((d, d)) / np.sqrt(d)
    for _ in range(5):
        D.pdf_dev(X, out)
        T.pdf_dev(X, out)
        D.reweight_dev(X, np.ones(d), F, out)
    torch.cuda.synchronize()
    D.close()
    T.close()
    del X, out
N, d = 1_000_000, 64
Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
out = torch.empty(N, d, dtype=torch.float64, device="cuda")
Q = 0.3 * np.eye(d) + 0.05 * np.random.default_rng(0).standard_normal((d, d))
for i in range(5):
    cusmc_amd.api.propagate_dev(Xp, anc, 0.9 * np.eye(d) + 0.01 * Q, Q, out, "mvn", 0.0, 1.0, seed=1, step=i + 1, ctx=ctx)
    cusmc_amd.api.propagate_dev(Xp, anc, 0.9 * np.eye(d), Q, out, "mvn", 0.0, 1.0, seed=1, step=i + 1, ctx=ctx)
    cusmc_amd.api.propagate_dev(Xp, anc, 0.9 * np.eye(d) + 0.01 * Q, np.tril(Q), out, "mvn", 0.0, 1.0, seed=1, step=i + 1, ctx=ctx)  # (TRIQ)
torch.cuda.synchronize()
# the wide proposal kernel (128 < d <= 256): dense and diagonal G, Normal and Student-t
for N, d in ((250_000, 256), (250_000, 192), (250_000, 144)):
    rng = np.random.default_rng(d)
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    Gm = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    Q = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    for i in range(3):
        for kind, nu in (("mvn", 0.0), ("mvt", 4.0)):
            cusmc_amd.api.propagate_dev(Xp, anc, Gm, Q, out, kind, nu, 1.0, seed=1, step=i + 1, ctx=ctx)
            cusmc_amd.api.propagate_dev(Xp, anc, np.diag(np.diag(Gm)), Q, out, kind, nu, 1.0, seed=1, step=i + 1, ctx=ctx)
            cusmc_amd.api.propagate_dev(Xp, anc, Gm, np.tril(Q), out, kind, nu, 1.0, seed=1, step=i + 1, ctx=ctx)  # (TRIQ)
    torch.cuda.synchronize()
    del Xp, anc, out
