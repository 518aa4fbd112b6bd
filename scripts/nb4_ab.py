"""A/B of the headline shape (1e6 x 64 MVN log-density): the hand-written assembly kernel (kernels/logpdf_nb4_gfx950.s)
against the compiled one (CUSMC_NB4_ASM=0), alternating child processes on one box: bit-identity of the outputs at
several N (incl. ragged last tiles) and launch time (HIP events, 600 warm-up launches, median of 5 x 400).
    python scripts/nb4_ab.py [rounds] [extra env as K=V ...]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import cusmc_amd
    import bench
    torch.cuda.set_device(0)
    D = 64
    mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), bench.make_sigma(D, 1))
    mvn.ctx.use_torch_stream()
    g = torch.Generator(device="cuda").manual_seed(1234)
    X = torch.randn(1_000_000, D, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(1_000_000, dtype=torch.float64, device="cuda")
    sums = []
    for n in (1_000_000, 999_999, 524_288 + 5, 300_001, 262_144 + 17):
        o = out[:n]
        o.fill_(float("nan"))
        mvn.pdf_dev(X[:n], o)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(o).all()), n
        sums.append(hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16])
    for _ in range(600):
        mvn.pdf_dev(X, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400):
            mvn.pdf_dev(X, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 400 * 1e3)
    ts.sort()
    print("RESULT %.2f %.2f %.2f %s" % (ts[2], ts[0], ts[-1], ",".join(sums)), flush=True)


def main():
    if os.environ.get("NB4_AB_CHILD"):
        return child()
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    extra = dict(kv.split("=", 1) for kv in sys.argv[2:])
    modes = ("0", "2", "1") if os.environ.get("NB4_AB_THREE") else ("0", "1")
    res = {m: [] for m in modes}
    sums = {}
    for r in range(rounds):
        for mode in modes:
            env = dict(os.environ, NB4_AB_CHILD="1", CUSMC_NB4_ASM=mode, **(extra if mode != "0" else {}))
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print("mode", mode, "FAILED:", out.stderr[-2000:])
                return 1
            med, lo, hi, s = line[0].split()[1:]
            res[mode].append(float(med))
            sums.setdefault(mode, s)
            assert sums[mode] == s, "outputs differ between runs of one mode"
            print("round %d %s: median %.2f us (min %.2f max %.2f)" % (r, {"0": "compiled   ", "1": "asm        ", "2": "asm BIRTH=0"}[mode], float(med), float(lo), float(hi)), flush=True)
    same = all(sums[m] == sums["0"] for m in modes)
    print("bit-identical outputs at all sizes:", same)
    mid = lambda v: sorted(v)[len(v) // 2]
    print("median of medians: " + " | ".join("%s %.2f us" % ({"0": "compiled", "1": "asm", "2": "asm BIRTH=0"}[m], mid(res[m])) for m in modes), extra or "")
    return 0 if same else 2


if __name__ == "__main__":
    sys.exit(main())
