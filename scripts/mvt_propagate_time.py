import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for d in (64, 8, 2):
    N = 1_000_000
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    for dense in (False, True):
        G = 0.9 * np.eye(d) + (0.01 * np.random.default_rng(1).standard_normal((d, d)) if dense else 0.0)
        Q = 0.3 * np.eye(d) + (0.05 * np.random.default_rng(0).standard_normal((d, d)) if dense else 0.0)
        st = [0]
        def f():
            st[0] += 1
            cusmc_amd.api.propagate_dev(Xp, anc, G, Q, out, "mvt", 4.0, 1.0, seed=1, step=st[0], ctx=ctx)
        print("propagate MVT d=%d %s: %.1f us" % (d, "dense" if dense else "diagonal", timed(f, 5, 2)), flush=True)
