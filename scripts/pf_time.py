import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd, torch
cusmc_amd.set_seed(1)
for (N, d, T) in [(1_000_000, 2, 100), (1_000_000, 8, 20), (200_000, 64, 10)]:
    I = np.eye(d)
    rng = np.random.default_rng(0)
    Y = np.cumsum(0.03 * rng.standard_normal((d, T)), axis=1)
    cusmc_amd.run(min(N, 1000), d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)  # load the code objects
    t0 = time.perf_counter()
    out = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.001 * I if d == 2 else 0.5 * I, 0.001 * I if d == 2 else 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
    t1 = time.perf_counter()
    w = out["weights"]
    print("run N=%d d=%d T=%d: %.3f s total (incl. D2H of %.1f GB), weights finite %s, mean w[-1]=%.3e" % (N, d, T, t1 - t0, (out["posterior_x"].nbytes + w.nbytes) / 1e9, np.isfinite(w).all(), w[-1].mean()))
