"""Per-launch duration of the headline kernel as a function of how long the chip has been under this load:
blocks of 20 launches, one HIP event pair per block, from an idle chip (1 s of sleep), 4000 launches, three
repetitions.  Answers: where in a burst does bench.py's timed window (launches 601..620) sit?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cusmc_amd
N, D = 1_000_000, 64
rng = np.random.default_rng(1)
A = rng.standard_normal((D, D))
mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), A @ A.T / D + np.eye(D))
mvn.ctx.use_torch_stream()
X = torch.randn(N, D, dtype=torch.float64, device="cuda")
out = torch.empty(N, dtype=torch.float64, device="cuda")
for _ in range(50):
    mvn.pdf_dev(X, out)
torch.cuda.synchronize()
BL, NB = 20, 200
for rep in range(3):
    time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(NB + 1)]
    ev[0].record()
    for b in range(NB):
        for _ in range(BL):
            mvn.pdf_dev(X, out)
        ev[b + 1].record()
    torch.cuda.synchronize()
    us = np.array([ev[b].elapsed_time(ev[b + 1]) / BL * 1e3 for b in range(NB)])
    idx = [0, 1, 2, 4, 7, 10, 15, 20, 25, 30, 40, 50, 75, 100, 150, 199]
    print("rep %d: " % rep + " ".join("%d:%.1f" % (i * BL, us[i]) for i in idx), flush=True)
    print("        min %.1f us at launch %d; mean of launches 600..619: %.1f; 2000..3999: %.1f" %
          (us.min(), int(us.argmin()) * BL, us[30], us[100:].mean()), flush=True)
