import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
ctx = cusmc_amd.api.default_context().use_torch_stream()
for N, d in ((500_000, 256), (1_000_000, 128), (1_000_000, 64)):
    rng = np.random.default_rng(1); A = rng.standard_normal((d, d)); S = A @ A.T / d + np.eye(d)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda"); out = torch.empty(N, dtype=torch.float64, device="cuda")
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), S, ctx=ctx)
    for _ in range(40): D.pdf_dev(X, out)
    torch.cuda.synchronize(); D.close()
