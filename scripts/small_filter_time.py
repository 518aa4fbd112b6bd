"""run() wall time for small filters (launch- and upload-bound regime).  Developer aid."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
cusmc_amd.set_seed(1)
for (N, d, T) in [(1000, 2, 100), (10_000, 2, 100), (100_000, 2, 100), (1000, 8, 100), (1000, 32, 100)]:
    I = np.eye(d)
    Y = np.cumsum(0.03 * np.random.default_rng(0).standard_normal((d, T)), axis=1)
    cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
        ts.append(time.perf_counter() - t0)  # (result kept alive: freeing it is not part of run())
        del res
    t = min(ts)
    print("run N=%d d=%d T=%d: %.2f ms total, %.1f us per time step" % (N, d, T, t * 1e3, t / (T - 1) * 1e6))
