"""Log-pdf timings at dimensions that are NOT multiples of 16 (padded MFMA variants; d < 16 generic).
Developer aid:  python scripts/generic_sweep.py"""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd, timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for d in (2, 8, 24, 40, 65, 100, 144, 200):
    N = 64_000_000 // d
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
    t = timed(lambda: D.pdf_dev(X, out), 50, 50)
    print("d=%d N=%d mvn %.1f us (%.2f TB/s algorithmic)" % (d, N, t, N * (8 * d + 8) / t / 1e6))
    D.close(); del X, out
