import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd, timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for d in (96, 112, 128):
    N = 1_000_000
    buf = torch.randn(N * d + 2, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
    for off, name in ((0, "aligned"), (1, "8-byte offset (padded tile kernel)")):
        X = buf[off:off + N * d].view(N, d)
        t = timed(lambda: D.pdf_dev(X, out), 50, 100)
        nb = d // 16
        print("d=%d %s: %.1f us, %.1f TFLOP/s" % (d, name, t, N / 16 * 2 * nb * (nb + 1) * 2048 / t / 1e6))
    D.close()
