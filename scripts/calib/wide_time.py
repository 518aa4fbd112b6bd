"""Times the wide log-pdf kernel per block count (d = 16 NB, NB = 12 .. 16) for the library in place."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cusmc_amd import _lib
if os.environ.get("EXP"):  # (A/B against a calibration build: see scripts/d128_ab.py)
    _lib.SO_PATH = _lib.SO_PATH.replace("libcusmc_hip.so", "libcusmc_hip_exp.so")
import numpy as np, torch
import cusmc_amd
from scripts.logpdf_sweep import timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
rng = np.random.default_rng(0)
for d in [int(v) for v in os.environ.get("WIDE_TIME_DS", "192,185,208,224,240,256").split(",")]:
    N = int(1.28e8 // d)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    A = rng.standard_normal((d, d))
    if os.environ.get("WIDE_TIME_NU"):  # Student-t instead (the epilogue's log1p)
        D = cusmc_amd.MultiVariateTStudentDistribution(rng.standard_normal(d), A @ A.T / d + np.eye(d), float(os.environ["WIDE_TIME_NU"]), ctx=ctx)
    else:
        D = cusmc_amd.MultiVariateNormalDistribution(rng.standard_normal(d), A @ A.T / d + np.eye(d), ctx=ctx)
    t = timed(lambda: D.pdf_dev(X, out), 20, 5)
    nb = (d + 15) // 16
    flop = 2.0 * nb * (nb + 1) * 2048 / 16
    print("%s d=%d N=%d: %.1f us, %.1f TFLOP/s" % (sys.argv[1] if len(sys.argv) > 1 else "", d, N, t, N * flop / t / 1e6), flush=True)
    D.close()
