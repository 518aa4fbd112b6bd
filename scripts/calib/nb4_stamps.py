"""Per-wave timeline of the assembly log-pdf kernel (kernels/logpdf_nb4_gfx950.s): entry and exit of each of the 2048
waves of one 1e6 x 64 launch (s_memrealtime, 100 MHz), after a warm-up; percentiles of the exit times and the mean exit
per workgroup class b % 8.   CUSMC_NB4_POOL_ROUNDS=<r> python scripts/calib/nb4_stamps.py"""
import os, sys, tempfile
path = os.path.join(tempfile.gettempdir(), "nb4_stamps.bin")
os.environ["CUSMC_NB4_STAMPS"] = path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, cusmc_amd, bench
D = 64
mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), bench.make_sigma(D, 1)); mvn.ctx.use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(1234)
X = torch.randn(1_000_000, D, dtype=torch.float64, device="cuda", generator=g); out = torch.empty(1_000_000, dtype=torch.float64, device="cuda")
for _ in range(600): mvn.pdf_dev(X, out)
ends, tots = [], []
for rep in range(20):
    for _ in range(20): mvn.pdf_dev(X, out)
    mvn.ctx.synchronize()
    st = np.fromfile(path, dtype=np.uint64).reshape(-1, 2)[:2048].astype(np.int64)
    t0 = st[:, 0].min()
    e = (st[:, 1] - t0) / 100.0  # us
    ends.append(np.sort(e)); tots.append(e.max())
    if rep == 0:
        print("entry spread %.2f us" % ((st[:, 0].max() - t0) / 100.0))
        print("exit by b %% 8:", np.round([e.reshape(256, 8)[b::8].mean() for b in range(8)], 2))
        print("exit by wave w:", np.round(e.reshape(256, 8).mean(0), 2))
E = np.mean(ends, 0)
print("pool rounds %s: wave exit (mean over 20 launches of the sorted exits) min %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f mean %.2f | launch end mean %.2f" % (
    os.environ.get("CUSMC_NB4_POOL_ROUNDS", "default"), E[0], E[204], E[1024], E[1843], E[2027], E[-1], E.mean(), np.mean(tots)))
