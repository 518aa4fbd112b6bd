"""Proposal kernels, Student-t against Normal, per nu (RNG contract 2: nu = 2, 4 closed form, others Marsaglia-Tsang).
    python scripts/mvt_nu_time.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = ((1_000_000, 64), (500_000, 256), (4_000_000, 8), (8_000_000, 2))
if os.environ.get("MVT_NU_SHAPES"):  # e.g. "1000000x64,500000x256"
    shapes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["MVT_NU_SHAPES"].split(","))
kinds = (("mvn", 0.0), ("mvt", 4.0), ("mvt", 2.0), ("mvt", 3.0), ("mvt", 30.0), ("mvt", 1.5))
if os.environ.get("MVT_NU_KINDS"):  # e.g. "0,4,3": 0 = Normal
    kinds = tuple(("mvn", 0.0) if float(v) == 0 else ("mvt", float(v)) for v in os.environ["MVT_NU_KINDS"].split(","))
for N, d in shapes:
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    Gd = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    Qd = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    st = [0]
    forms = [("diag G, diag Q", np.diag(np.diag(Gd)), np.diag(np.diag(Qd))), ("dense G, dense Q", Gd, Qd)]
    if d >= 32:  # a lower triangular Q (Cholesky factor) takes the triangular instantiations
        forms.append(("dense G, lower Q", Gd, np.tril(Qd)))
    for label, Gm, Qm in forms:
        base = None
        for kind, nu in kinds:
            def f():
                st[0] += 1
                cusmc_amd.api.propagate_dev(Xp, anc, Gm, Qm, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
            t = timed(f, 8, 3)
            base = base or t
            print("%s N=%d d=%d %s %s nu=%g: %.1f us  (%.2f x Normal)" % (tag, N, d, label, kind, nu, t, t / base), flush=True)
    del Xp, anc, out
