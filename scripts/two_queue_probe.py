"""Does a fused filter step split into two half-size launches on TWO streams of one device cost what one launch
costs?  (What the devices=[0,0] rehearsal of the sharded loop can show at best.)  Unsharded kernel, no cross-waits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cusmc_amd
from cusmc_amd import api
N, d, B = 1_000_000, 2, 10
I = np.eye(d)
g = torch.Generator(device="cuda").manual_seed(1)
wp = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
Xp = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
a = torch.empty(N, dtype=torch.int32, device="cuda"); X = torch.empty(N, d, dtype=torch.float64, device="cuda"); w = torch.empty(N, dtype=torch.float64, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs, obs = [], []
for s in streams:
    c = cusmc_amd.Context(); c.set_stream(s.cuda_stream); ctxs.append(c)
    obs.append(cusmc_amd.MultiVariateNormalDistribution(None, 0.5 * I, ctx=c))
y = np.zeros(d)
def step(parts, st):
    n = N // parts
    for r in range(parts):
        f = r * n
        api.pf_step_dev(obs[r % 2], wp, Xp, 0.95 * I, 0.3 * I, y, None, a[f:f + n], X[f:f + n], w[f:f + n], B=B, seed=1, step=st, first=f)
for parts in (1, 2, 4):
    for _ in range(5): step(parts, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for s in streams: s.wait_event(e0)
    for it in range(50): step(parts, it + 2)
    for s in streams: e1.wait(s) if False else torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    print("parts=%d on %d streams: %.1f us per step" % (parts, min(parts, 2), e0.elapsed_time(e1) / 50 * 1e3), flush=True)
