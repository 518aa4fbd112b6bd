"""Shifted-form (mu != 0) log-pdf timings over d for A/B between library builds.  Developer aid."""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd, timed
tag = sys.argv[1]
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
res = []
for d in (16, 32, 48, 64):
    N = 64_000_000 // d
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    for kind in ("mvn+mu", "mvt+mu"):
        D = (cusmc_amd.MultiVariateTStudentDistribution(np.full(d, 0.1), spd(d, 1), 4.0, ctx=ctx) if kind == "mvt+mu"
             else cusmc_amd.MultiVariateNormalDistribution(np.full(d, 0.1), spd(d, 1), ctx=ctx))
        res.append("d=%d %s %.1f" % (d, kind, timed(lambda: D.pdf_dev(X, out), 200, 300)))
        D.close()
    del X, out
print(tag, " | ".join(res))
