"""Phase timing of cusmc_pf_run_host (CUSMC_TRACE=1 prints the phases to stderr).  Developer aid."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
for (N, d, T) in [(1_000_000, 2, 100), (1_000_000, 8, 20), (200_000, 64, 10)]:
    I = np.eye(d)
    Y = np.cumsum(0.03 * np.random.default_rng(0).standard_normal((d, T)), axis=1)
    for i in range(2):
        t0 = time.perf_counter()
        out = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
        print("N=%d d=%d T=%d python-level total %.1f ms" % (N, d, T, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
        del out
