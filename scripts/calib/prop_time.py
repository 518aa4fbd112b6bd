"""Times the proposal kernels at 1e6 x 64 (diagonal and dense G and Q) and 5e5 x 256 for the library in place.
Developer aid for ablation builds (e.g. -DCUSMC_ABL_NO_RNG)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for N, d in ((1_000_000, 64), (500_000, 256)):
    Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
    ident = torch.arange(N, dtype=torch.int32, device="cuda")
    out = torch.empty(N, d, dtype=torch.float64, device="cuda")
    Gd = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    Qd = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    st = [0]
    for label, Gm, Qm in (("diag G, diag Q", np.diag(np.diag(Gd)), np.diag(np.diag(Qd))), ("diag G, dense Q", np.diag(np.diag(Gd)), Qd), ("dense G, dense Q", Gd, Qd)):
        for kind, nu in (("mvn", 0.0), ("mvt", 4.0)):
            for aname, av in (("random ancestors", anc), ("identity ancestors", ident)):
                def f():
                    st[0] += 1
                    cusmc_amd.api.propagate_dev(Xp, av, Gm, Qm, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
                print("%s N=%d d=%d %s %s, %s: %.1f us" % (tag, N, d, label, kind, aname, timed(f, 5, 2)), flush=True)
    del Xp, anc, out
