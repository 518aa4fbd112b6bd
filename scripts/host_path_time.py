"""PCIe-inclusive rate of the host-pointer entry points (what the R-level API pays).  Developer aid."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd
N, d = 1_000_000, 64
X = np.random.default_rng(0).standard_normal((N, d))
D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1))
D.pdf_batch(X[:1000])
for rep in range(3):
    t0 = time.perf_counter()
    out = D.pdf_batch(X)
    dt = time.perf_counter() - t0
    print("cusmc_dist_pdf_host N=%d d=%d: %.1f ms = %.3g evals/s = %.1f GB/s of X" % (N, d, dt * 1e3, N / dt, N * d * 8 / dt / 1e9))
D.close()
