import sys, numpy as np, subprocess, os
sys.path.insert(0, '.')
mode = sys.argv[1] if len(sys.argv) > 1 else "parent"
if mode == "parent":
    for m in ("ctx_then_torch", "torch_then_ctx", "ctx_pdf_then_torch"):
        r = subprocess.run([sys.executable, __file__, m], capture_output=True, text=True)
        print(m, "->", (r.stdout.strip().splitlines() or ["<no stdout>"])[-1], "| rc", r.returncode, "|", (r.stderr.strip().splitlines() or [""])[-1][:120])
    sys.exit(0)
import cusmc_amd
if mode == "ctx_then_torch":
    ctx = cusmc_amd.Context()
    import torch
    x = torch.zeros(4, device="cuda"); print("ok", x.sum().item())
elif mode == "torch_then_ctx":
    import torch
    x = torch.zeros(4, device="cuda")
    ctx = cusmc_amd.Context(); print("ok")
elif mode == "ctx_pdf_then_torch":
    ctx = cusmc_amd.Context()
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(64), np.eye(64), ctx=ctx)
    D.pdf_batch(np.zeros((40, 64)))
    import torch
    x = torch.zeros(4, device="cuda"); print("ok", x.sum().item())
