#!/usr/bin/env python3
"""Phase timing of run() for a small filter (CUSMC_TRACE=1 prints the phases of cusmc_pf_run_host).
    CUSMC_TRACE=1 python scripts/run_trace_small.py [N] [T]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = 2
I = np.eye(d)
Y = np.random.default_rng(0).standard_normal((d, T))
for rep in range(4):
    t0 = time.perf_counter()
    res = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
    print("rep %d: %.3f ms" % (rep, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    del res
