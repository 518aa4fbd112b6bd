#!/usr/bin/env python3
"""A/B between the library and a calibration build of it (libcusmc_hip_exp.so, e.g. built with
-DEXP_NOLDSW: factor fragments read once per tile, wrong results) on the LDS-factor shapes:
    make -C cusmc_amd/csrc OUT=../libcusmc_hip_exp.so OBJDIR=build_exp CXXFLAGS="<library flags> -DEXP_NOLDSW"
    python scripts/d128_ab.py; EXP=nolds python scripts/d128_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cusmc_amd import _lib
if os.environ.get("EXP"):
    _lib.SO_PATH = _lib.SO_PATH.replace("libcusmc_hip.so", "libcusmc_hip_exp.so")
import numpy as np, torch, cusmc_amd
from scripts.logpdf_sweep import spd, timed
ctx = cusmc_amd.api.default_context().use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(7)
for d in tuple(int(v) for v in os.environ.get("DIMS", "80,96,112,128").split(",")):
    N = 1_000_000
    X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
    t = timed(lambda: D.pdf_dev(X, out), 50, 100)
    nb = d // 16
    print("%s d=%d: %.1f us, %.1f TFLOP/s" % (os.environ.get("EXP", "base"), d, t, N / 16 * 2 * nb * (nb + 1) * 2048 / t / 1e6), flush=True)
    D.close()
