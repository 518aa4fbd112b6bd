#!/usr/bin/env python3
"""Interpreter exit with live distribution handles (normally, and through an exception) must be clean:
the context closes its distributions first whatever order the globals are torn down in."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, cusmc_amd
D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(3), np.eye(3))
E = cusmc_amd.MultiVariateTStudentDistribution(np.zeros(70), np.eye(70), 4.0)
print(D.pdf_batch(np.zeros((4, 3)))[0], cusmc_amd.MVNPDF(np.zeros(2), np.zeros(2), np.eye(2)))
keep = [D, E]
if len(sys.argv) > 1:
    raise RuntimeError("exit through an exception with live handles")
