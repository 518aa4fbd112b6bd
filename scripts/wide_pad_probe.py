"""PAD code path against row stride: the same d from an aligned pointer, an 8-byte-offset pointer (PAD variant,
nothing to mask) and a row stride of d + 2 (plain variant).  Developer aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cusmc_amd
from scripts.logpdf_sweep import spd, timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for d in tuple(int(v) for v in os.environ.get("DIMS", "192,256").split(",")):
    N = 64_000_000 // d
    buf = torch.randn(N * (d + 2) + 2, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
    for name, X in (("aligned", buf[:N * d].view(N, d)), ("8-byte offset (PAD path, same d)", buf[1:1 + N * d].view(N, d)),
                    ("row stride d+2, aligned (plain path)", buf[:N * (d + 2)].view(N, d + 2)[:, :d])):
        t = timed(lambda: D.pdf_dev(X, out), 50, 50)
        print("d=%d %s: %.1f us" % (d, name, t), flush=True)
    D.close()
