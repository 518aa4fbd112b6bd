"""Latency of the R-level scalar calls (one density / one draw / one small resample).  Developer aid."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from scripts.logpdf_sweep import spd
for d in (2, 8, 64, 256):
    S, mu, x = spd(d, 1), np.zeros(d), np.ones(d) * 0.1
    cusmc_amd.MVNPDF(x, mu, S); cusmc_amd.MVTPDF(x, mu, S, 4.0); cusmc_amd.MVN(mu, S)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): cusmc_amd.MVNPDF(x, mu, S)
    t1 = time.perf_counter()
    for _ in range(n): cusmc_amd.MVTPDF(x, mu, S, 4.0)
    t2 = time.perf_counter()
    for _ in range(n): cusmc_amd.MVN(mu, S)
    t3 = time.perf_counter()
    print("d=%d: MVNPDF %.0f us, MVTPDF %.0f us, MVN draw %.0f us per call" % (d, (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, (t3 - t2) / n * 1e6))
w = np.random.default_rng(0).random(1000)
cusmc_amd.metropolis_hastings(w, 1000, 10)
t0 = time.perf_counter()
for _ in range(200): cusmc_amd.metropolis_hastings(w, 1000, 10)
print("metropolis_hastings(N=1000, B=10): %.0f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
