"""Times the d = 256 dense proposal (5e5 particles, dense G and Q) for the library currently in place.
Developer aid for the ablation builds of kernels/propagate_mfma_wide.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
d, N = 256, 500_000
Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
out = torch.empty(N, d, dtype=torch.float64, device="cuda")
G = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
Q = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
st = [0]
for kind, nu in (("mvn", 0.0), ("mvt", 4.0)):
    def f():
        st[0] += 1
        cusmc_amd.api.propagate_dev(Xp, anc, G, Q, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
    print("%s %s: %.1f us" % (sys.argv[1] if len(sys.argv) > 1 else "", kind, timed(f, 5, 2)), flush=True)
