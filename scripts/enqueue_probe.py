import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, cusmc_amd, bench
from cusmc_amd import sharding
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1234)
N, D = bench.N_PER_GPU, bench.D
X = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g); out = torch.empty(N, dtype=torch.float64, device="cuda")
mvn = cusmc_amd.MultiVariateNormalDistribution(np.zeros(D), bench.make_sigma(D, 1)); mvn.ctx.use_torch_stream()
for W in (1, 8, 16, 64):
    cnt = N // W
    Xs, outs = X[:cnt], out[:cnt]
    for _ in range(600): mvn.pdf_dev(Xs, outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(400): mvn.pdf_dev(Xs, outs)
    e1.record(); t_enq = time.perf_counter() - t
    torch.cuda.synchronize(); t_all = time.perf_counter() - t
    print("W=%d cnt=%d: host enqueue %.1f us/call, wall %.1f us/call, events %.1f us/call" % (W, cnt, t_enq / 400 * 1e6, t_all / 400 * 1e6, e0.elapsed_time(e1) / 400 * 1e3), flush=True)
