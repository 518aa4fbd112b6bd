"""What ONE GPU can say about bench.py's multi-GPU legs: each rank's share at world = 1, 2, 4, 8, launched exactly
as bench.py launches it (same Python call, same wall-clock timing around a loop of launches), back to back on
one device.  The legs have no data-path collective, so the predicted strong-scaling efficiency of the driver's
curve is  t(1) / (W * t(W-share))  -- host enqueue, launch gap, prologue and tail included.
    python scripts/scaling_prediction.py > profiles/r03_scaling_prediction.md"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402
from cusmc_amd import sharding  # noqa: E402
import bench  # noqa: E402


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(1234)
    N, D = bench.N_PER_GPU, bench.D
    X = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), bench.make_sigma(D, 1))
    mvn.ctx.use_torch_stream()
    print("# Scaling prediction from one MI355X (round 3)\n")
    print("`scripts/scaling_prediction.py`: every rank's share of bench.py's fixed-total legs, timed on one GPU the way "
          "bench.py times it (wall clock around a loop of launches from Python, 600 warm-up launches).  No collective sits "
          "in these legs, so the driver's N-GPU value should be  total / t(share)  and its efficiency  t(1) / (W t(share)).\n")
    print("## `strong.headline_1e6x64`: 1e6 x 64 log-densities split over W ranks\n")
    print("| W | particles per rank | us per launch (wall) | evals/s over W ranks | speed-up | efficiency |")
    print("|---|---|---|---|---|---|")
    t1 = None
    for W in (1, 2, 4, 8):
        _, cnt = sharding.shard_range(N, 0, W)
        Xs, outs = X[:cnt], out[:cnt]
        t = timed(lambda: mvn.pdf_dev(Xs, outs), 400, 600)
        t1 = t1 or t
        print("| %d | %d | %.1f | %.3g | %.2f | %.0f %% |" % (W, cnt, t * 1e6, N / t, t1 / t, 100 * t1 / t / W), flush=True)
    print("\n## `strong.c5_4e6x256`: 4e6 x 256 split over W ranks (a one-GPU box holds W >= 2 shares: 8.2 GB at W = 1 fits too)\n")
    print("| W | particles per rank | ms per pass | TFLOP/s per GPU | speed-up | efficiency |")
    print("|---|---|---|---|---|---|")
    d256 = cusmc_amd.MultiVariateNormalDistribution(np.zeros(bench.C5_D), bench.make_sigma(bench.C5_D, 5))
    d256.ctx.use_torch_stream()
    nb = bench.C5_D // 16
    flop5 = 2.0 * nb * (nb + 1) * 2048 / 16
    t1 = None
    for W in (1, 2, 4, 8):
        _, cnt = sharding.shard_range(bench.C5_N, 0, W)
        X5 = torch.randn(cnt, bench.C5_D, dtype=torch.float64, device="cuda", generator=g)
        o5 = torch.empty(cnt, dtype=torch.float64, device="cuda")
        t = timed(lambda: d256.pdf_dev(X5, o5), 10, 3)
        t1 = t1 or t
        print("| %d | %d | %.3f | %.1f | %.2f | %.0f %% |" % (W, cnt, t * 1e3, cnt * flop5 / t / 1e12, t1 / t, 100 * t1 / t / W), flush=True)
        del X5, o5
    print("\n## `mh`: configs[1]'s resampler, weak (every rank 1e5 chains x B = 1e3 over a W x 1e5 weight vector); "
          "the all-gather of the W x 0.8 MB weight shards is not in this number\n")
    print("| W | weight vector | ms per resample (one rank's chains) | steps/s over W ranks | per-rank rate vs W = 1 |")
    print("|---|---|---|---|---|")
    a = torch.empty(bench.MH_N, dtype=torch.int32, device="cuda")
    st = [0]
    r1 = None
    for W in (1, 2, 4, 8):
        w = torch.rand(W * bench.MH_N, dtype=torch.float64, device="cuda", generator=g) * 1e-20

        def f():
            st[0] += 1
            cusmc_amd.Sampler.metropolis_hastings_dev(w, a, B=bench.MH_B, t=st[0], seed=1, first=0, ctx=mvn.ctx)
        t = timed(f, 5, 2)
        rate = bench.MH_N * bench.MH_B / t
        r1 = r1 or rate
        print("| %d | %d | %.3f | %.3g | %.2f |" % (W, W * bench.MH_N, t * 1e3, W * rate, rate / r1), flush=True)
    print("\nThe headline itself is weak-scaled (every rank its own 1e6 x 64, no collective): its efficiency is 1 by "
          "construction up to the node's power and host scheduling, which one GPU cannot show.")


if __name__ == "__main__":
    main()
