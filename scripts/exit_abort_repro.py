#!/usr/bin/env python3
"""Round 1 saw `terminate called after throwing an instance of 'std::bad_variant_access'` when pytest
exited after a FAILED gpu test (gpurun_out/pytest.txt).  This reproduces the situation outside pytest: live
distribution handles on torch's stream, device tensors, pending work, and an exit through an exception /
sys.exit / os._exit-free normal return.  CUSMC_TRACE_TERMINATE=1 prints the stack that reached
std::terminate.  usage: exit_abort_repro.py {normal|raise|sysexit|keepframe}"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cusmc_amd
from cusmc_amd import api

mode = sys.argv[1] if len(sys.argv) > 1 else "normal"
torch.cuda.set_device(0)
obs = cusmc_amd.MultiVariateTStudentDistribution(None, np.eye(2), 4.0)
obs.ctx.use_torch_stream()
N = 20000
wp = torch.rand(N, dtype=torch.float64, device="cuda")
Xp = torch.randn(N, 2, dtype=torch.float64, device="cuda")
a = torch.empty(N, dtype=torch.int32, device="cuda")
X1 = torch.empty(N, 2, dtype=torch.float64, device="cuda")
w1 = torch.empty(N, dtype=torch.float64, device="cuda")
api.pf_step_dev(obs, wp, Xp, np.eye(2), np.eye(2), np.zeros(2), None, a, X1, w1, kind="mvt", nu=4.0)
other = cusmc_amd.MultiVariateNormalDistribution(np.zeros(70), np.eye(70))
print("work enqueued:", float(w1.sum()), flush=True)
kept = []


def fail_like_a_test():
    local_obs = cusmc_amd.MultiVariateNormalDistribution(np.zeros(3), np.eye(3))
    t = torch.ones(5, device="cuda")
    assert torch.equal(t, t + 1), "a failing comparison with live handles in the frame"


if mode == "raise":
    fail_like_a_test()
elif mode == "keepframe":  # what pytest does: keep the traceback (and with it the frame's locals) until exit
    try:
        fail_like_a_test()
    except AssertionError:
        kept.append(sys.exc_info())
    sys.exit(1)
elif mode == "sysexit":
    sys.exit(3)
