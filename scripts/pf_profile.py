import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
cusmc_amd.set_seed(1)
N, d, T = 1_000_000, 2, 20
I = np.eye(d)
Y = np.cumsum(0.03 * np.random.default_rng(0).standard_normal((d, T)), axis=1)
out = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.001 * I, 0.001 * I, 0.0, "metropolis", "mvn", seed=3)
N, d, T = 200_000, 64, 6
I = np.eye(d)
Y = np.cumsum(0.03 * np.random.default_rng(0).standard_normal((d, T)), axis=1)
out = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
