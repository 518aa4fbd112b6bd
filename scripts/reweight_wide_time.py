import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd, timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for d in tuple(int(v) for v in os.environ.get("DIMS", "128,144,160,176,192,200,224,256").split(",")):
    N = 64_000_000 // d
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
    F = np.eye(d) + 0.01 * np.random.default_rng(2).standard_normal((d, d))
    y = np.ones(d)
    t0 = timed(lambda: D.pdf_dev(X, out), 50, 50)
    t1 = timed(lambda: D.reweight_dev(X, y, F, out), 50, 50)
    print("d=%d N=%d pdf %.1f us reweight(F dense) %.1f us" % (d, N, t0, t1), flush=True)
