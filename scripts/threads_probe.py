import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd, timed
tag = sys.argv[1]
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
res = []
for d in (80, 96, 112, 128, 100):
    N = 64_000_000 // d
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    for kind in ("mvn", "mvt", "reweight"):
        D = (cusmc_amd.MultiVariateTStudentDistribution(np.zeros(d), spd(d, 1), 4.0, ctx=ctx) if kind == "mvt"
             else cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx))
        if kind == "reweight":
            F = np.eye(d) + 0.01 * np.random.default_rng(2).standard_normal((d, d)); y = np.zeros(d)
            fn = lambda: D.reweight_dev(X, y, F, out)
        else:
            fn = lambda: D.pdf_dev(X, out)
        res.append("d=%d %s %.1f" % (d, kind, timed(fn, 100, 200)))
        D.close()
    del X, out
print(tag, " | ".join(res))
