"""Do consecutive launches of the log-pdf kernel on one stream OVERLAP (launch n + 1's first waves entering before launch
n's last wave has left)?  Per-wave s_memrealtime stamps (100 MHz, chip-wide) of the last two of a train of launches."""
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
for rep in range(5):
    for _ in range(40): mvn.pdf_dev(X, out)
    mvn.ctx.synchronize()
    st = np.fromfile(path, dtype=np.uint64).reshape(2, -1, 2)[:, :2048].astype(np.int64)
    a, b = (st[0], st[1]) if st[0][:, 0].min() < st[1][:, 0].min() else (st[1], st[0])   # a = the earlier launch
    t0 = a[:, 0].min()
    print("launch n: first entry 0, last entry %.2f, first exit %.2f, last exit %.2f | launch n+1: first entry %.2f, last entry %.2f, last exit %.2f  (us)  => period %.2f, gap last-exit -> first-entry %.2f" % (
        (a[:, 0].max() - t0) / 100, (a[:, 1].min() - t0) / 100, (a[:, 1].max() - t0) / 100, (b[:, 0].min() - t0) / 100, (b[:, 0].max() - t0) / 100,
        (b[:, 1].max() - t0) / 100, (b[:, 0].min() - t0) / 100, (b[:, 0].min() - a[:, 1].max()) / 100))
