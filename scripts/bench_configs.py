#!/usr/bin/env python3
"""Times the per-GPU share of every BASELINE.json config on one MI355X (device-resident inputs,
HIP events on the launch stream) and prints a markdown table for BASELINE.md section 4.
Not the judged bench (that is bench.py); these are the parity-test shapes, timed."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402


def spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


def timed(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    rows = []
    ctx = cusmc_amd.api.default_context().use_torch_stream()
    g = torch.Generator(device="cuda").manual_seed(7)
    # log-pdf shapes
    for name, N, d, dist, nu in (("C1 MVNPDF 1e4 x d=8", 10_000, 8, "mvn", 0.0),
                                 ("headline MVN 1e6 x d=64", 1_000_000, 64, "mvn", 0.0),
                                 ("C4 MVT(nu=4) 1e6 x d=64", 1_000_000, 64, "mvt", 4.0),
                                 ("C5 share MVN 5e5 x d=256", 500_000, 256, "mvn", 0.0),
                                 ("MVN 1e6 x d=128", 1_000_000, 128, "mvn", 0.0),
                                 ("MVN 2e6 x d=32", 2_000_000, 32, "mvn", 0.0)):
        X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        out = torch.empty(N, dtype=torch.float64, device="cuda")
        D = (cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx) if dist == "mvn"
             else cusmc_amd.MultiVariateTStudentDistribution(np.zeros(d), spd(d, 1), nu, ctx=ctx))
        t = timed(lambda: D.pdf_dev(X, out), 400 if N * d <= 7e7 else 100, 400)  # (clock governor: ~50 ms to settle)
        nb = d // 16
        flops = N / 16 * 2 * nb * (nb + 1) * 2048 if d % 16 == 0 else N * (d * d + 4 * d)
        rows.append((name, "%.3g evals/s" % (N / t), "%.1f us" % (t * 1e6), "%.2f TB/s" % (N * (8 * d + 8) / t / 1e12),
                     "%.1f TFLOP/s" % (flops / t / 1e12)))
        D.close()
        del X, out
    # reweight_G with a general (dense) observation matrix F: QL-rotated on the host to the triangular form
    for name, N, d in (("reweight 1e6 x d=64, dense F", 1_000_000, 64), ("reweight 5e5 x d=256, dense F", 500_000, 256)):
        X = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        out = torch.empty(N, dtype=torch.float64, device="cuda")
        D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), spd(d, 1), ctx=ctx)
        F = np.eye(d) + 0.3 * np.random.default_rng(2).standard_normal((d, d)) / np.sqrt(d)
        y = np.ones(d)
        t = timed(lambda: D.reweight_dev(X, y, F, out), 200 if d == 64 else 50, 200)
        nb = d // 16
        rows.append((name, "%.3g evals/s" % (N / t), "%.1f us" % (t * 1e6), "%.2f TB/s" % (N * (8 * d + 8) / t / 1e12),
                     "%.1f TFLOP/s" % (N / 16 * 2 * nb * (nb + 1) * 2048 / t / 1e12)))
        D.close()
        del X, out
    # resampler shapes
    for name, N, B in (("C2 MH 1e5 chains x 1e3", 100_000, 1000), ("C3 step MH 1e6 x 10", 1_000_000, 10),
                       ("C4 share MH 1.25e5 of 1e6 chains x 1e4", 1_000_000, 10_000)):
        w = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 1e-20
        count = 125_000 if "share" in name else N
        a = torch.empty(count, dtype=torch.int32, device="cuda")
        st = [0]

        def f():
            st[0] += 1
            cusmc_amd.Sampler.metropolis_hastings_dev(w, a, B=B, t=st[0], seed=1, first=0, ctx=ctx)
        t = timed(f, 3 if B >= 1000 else 50, 1)
        rows.append((name, "%.3g steps/s" % (count * B / t), "%.3f ms" % (t * 1e3), "-", "-"))
    # proposal draws (propagate_K), device-resident
    for name, N, d, kind, nu in (("propagate MVN 1e6 x d=64", 1_000_000, 64, "mvn", 0.0), ("propagate MVT(4) 1e6 x d=64", 1_000_000, 64, "mvt", 4.0),
                                 ("propagate MVN 1e6 x d=2", 1_000_000, 2, "mvn", 0.0), ("propagate MVN 1e6 x d=8", 1_000_000, 8, "mvn", 0.0)):
        Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
        out = torch.empty(N, d, dtype=torch.float64, device="cuda")
        Gm = 0.9 * np.eye(d); Q = 0.3 * np.eye(d)
        st = [0]

        def f():
            st[0] += 1
            cusmc_amd.api.propagate_dev(Xp, anc, Gm, Q, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
        t = timed(f, 10, 2)
        rows.append((name, "%.3g particles/s" % (N / t), "%.1f us" % (t * 1e6), "%.2f TB/s" % (N * (16 * d + 4) / t / 1e12), "-"))
        del Xp, anc, out
    # dense proposal draws (eigen-sqrt Q and a dense G, as a filter with a general model runs them)
    # (and the same with Q lower triangular -- a Cholesky factor: Q xi over the k-blocks kb <= cb only)
    for name, N, d, kind, nu, lower in (("propagate dense MVN 1e6 x d=64", 1_000_000, 64, "mvn", 0.0, False),
                                        ("propagate dense MVT(4) 1e6 x d=64", 1_000_000, 64, "mvt", 4.0, False),
                                        ("propagate dense G, lower Q MVN 1e6 x d=64", 1_000_000, 64, "mvn", 0.0, True),
                                        ("propagate dense G, lower Q MVT(4) 1e6 x d=64", 1_000_000, 64, "mvt", 4.0, True),
                                        ("C5 share propagate dense MVN 5e5 x d=256", 500_000, 256, "mvn", 0.0, False),
                                        ("C5 share propagate dense MVT(4) 5e5 x d=256", 500_000, 256, "mvt", 4.0, False),
                                        ("C5 share propagate dense G, lower Q MVN 5e5 x d=256", 500_000, 256, "mvn", 0.0, True),
                                        ("C5 share propagate dense G, lower Q MVT(4) 5e5 x d=256", 500_000, 256, "mvt", 4.0, True)):
        rng = np.random.default_rng(d)
        Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        anc = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
        out = torch.empty(N, d, dtype=torch.float64, device="cuda")
        Gm = 0.9 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
        Q = 0.3 * np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
        if lower:
            Q = np.tril(Q)
        st = [0]

        def f():
            st[0] += 1
            cusmc_amd.api.propagate_dev(Xp, anc, Gm, Q, out, kind, nu, 1.0, seed=1, step=st[0], ctx=ctx)
        t = timed(f, 10, 2)
        flops = 2.0 * d * d + (2.0 * 16 * (d // 16) * (16 * (d // 16) + 16) / 2 if lower else 2.0 * d * d)  # (block-triangular count)
        rows.append((name, "%.3g particles/s" % (N / t), "%.1f us" % (t * 1e6), "%.2f TB/s" % (N * (16 * d + 4) / t / 1e12),
                     "%.1f TFLOP/s" % (N * flops / t / 1e12)))
        del Xp, anc, out
    # one filter time step (resample + propagate + reweight), device-resident: fused launch vs three
    for name, N, d in (("C3 filter step N=1e6 d=2 B=10", 1_000_000, 2), ("filter step N=1e6 d=8 B=10", 1_000_000, 8)):
        I = np.eye(d)
        Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        wp = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
        a = torch.empty(N, dtype=torch.int32, device="cuda")
        Xo = torch.empty(N, d, dtype=torch.float64, device="cuda")
        wo = torch.empty(N, dtype=torch.float64, device="cuda")
        obs = cusmc_amd.MultiVariateNormalDistribution(None, 0.5 * I, ctx=ctx)
        y = np.zeros(d)
        st = [0]

        def fused():
            st[0] += 1
            cusmc_amd.api.pf_step_dev(obs, wp, Xp, I, 0.3 * I, y, I, a, Xo, wo, B=10, seed=1, step=st[0])

        def three():
            st[0] += 1
            cusmc_amd.Sampler.metropolis_hastings_dev(wp, a, B=10, t=st[0], seed=1, ctx=ctx)
            cusmc_amd.api.propagate_dev(Xp, a, I, 0.3 * I, Xo, "mvn", 0.0, 1.0, seed=1, step=st[0], ctx=ctx)
            obs.reweight_dev(Xo, y, I, wo, log=False)
        tf, t3 = timed(fused, 50, 5), timed(three, 50, 5)
        rows.append((name, "%.3g particle-steps/s" % (N / tf), "%.1f us fused (%.1f us as three launches)" % (tf * 1e6, t3 * 1e6),
                     "%.2f TB/s" % (N * (16 * d + 12) / tf / 1e12), "-"))
        obs.close()
        del Xp, wp, a, Xo, wo
    # filter (host round trip included: that is what run() does)
    for name, N, d, T in (("C3 run() N=1e6 d=2 T=100", 1_000_000, 2, 100), ("run() N=2e5 d=64 T=10", 200_000, 64, 10)):
        I = np.eye(d)
        Y = np.cumsum(0.03 * np.random.default_rng(0).standard_normal((d, T)), axis=1)
        cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)  # (first call: HIP initialisation, code objects)
        t0 = time.perf_counter()
        res = cusmc_amd.run(N, d, T, Y, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, 0.0, "metropolis", "mvn", seed=3)
        t = time.perf_counter() - t0  # (the result stays alive: releasing 2.4 GB of pages costs 0.1 s by itself)
        del res
        rows.append((name, "%.3g particle-steps/s" % (N * (T - 1) / t), "%.3f s wall" % t, "-", "-"))
    # last column: the larger of the two roofline fractions -- algorithmic bytes against the 8 TB/s HBM spec, matrix-core
    # flops against the 78.6 TFLOP/s f64 MFMA peak (the kernels below ~0.3 on both are VALU- or latency-bound: DESIGN.md 4)
    print("| config | rate | time | algorithmic HBM | MFMA | of the nearer roofline |\n|---|---|---|---|---|---|")
    for r in rows:
        hbm = float(r[3].split()[0]) / 8.0 if r[3] != "-" else 0.0
        mfma = float(r[4].split()[0]) / 78.6 if r[4] != "-" else 0.0
        frac = "-" if max(hbm, mfma) == 0.0 else "%.0f %% (%s)" % (100 * max(hbm, mfma), "HBM" if hbm >= mfma else "MFMA")
        print("| " + " | ".join(r) + " | " + frac + " |")


if __name__ == "__main__":
    main()
