#!/usr/bin/env python3
"""Where does the first call of a process spend its time?  (HIP runtime initialisation + torch import in the
very first call; after that each .hip file's code object loads lazily at its first launch.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import cusmc_amd  # noqa: E402
print("import cusmc_amd (+ torch preload): %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)


def timed(label, fn):
    t = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    fn()
    t2 = time.perf_counter()
    print("%-42s first %.1f ms, second %.2f ms" % (label, (t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)


I2, I64 = np.eye(2), np.eye(64)
timed("context + MVNPDF d=2 (logpdf_generic.hip)", lambda: cusmc_amd.MVNPDF(np.zeros(2), np.zeros(2), I2))
timed("MVNPDF d=64 (logpdf_mfma.hip)", lambda: cusmc_amd.MVNPDF(np.zeros(64), np.zeros(64), I64))
timed("MVNPDF d=200 (logpdf_mfma_wide.hip)", lambda: cusmc_amd.MVNPDF(np.zeros(200), np.zeros(200), np.eye(200)))
timed("metropolis_hastings (resample.hip)", lambda: cusmc_amd.metropolis_hastings(np.ones(100), 100, 10))
timed("MVN draw d=2 (propagate.hip)", lambda: cusmc_amd.MVN(np.zeros(2), I2))
timed("MVN draw d=64 dense (propagate_mfma.hip)", lambda: cusmc_amd.MVN(np.zeros(64), I64 + 0.01))
Y = np.zeros((2, 5))
timed("run d=2 (pf_step.hip)", lambda: cusmc_amd.run(100, 2, 5, Y, np.zeros(2), I2, I2, I2, I2, I2, 0.0, "metropolis", "mvn", seed=1))
