"""The shader clock the chip holds under each kernel's load: launches are queued for ~4 s while `rocm-smi` is polled
from a child process (sysfs reads; the child never touches HIP).  Answers: how far below the boost clock do the
headline kernel and the d = 256 kernel run -- i.e. what is the matrix-core peak at the clock they actually get?"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cusmc_amd

def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10)
            sclk = re.findall(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", r.stdout)
            mclk = re.findall(r"mclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", r.stdout)
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", r.stdout)
            out.append((time.perf_counter(), sclk[:1], mclk[:1], pw[:1]))
        except Exception as e:  # noqa: BLE001
            out.append((time.perf_counter(), repr(e)))
        time.sleep(0.25)

ctx = cusmc_amd.api.default_context().use_torch_stream()
rng = np.random.default_rng(1)
r0 = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
print("idle:\n" + "\n".join(l for l in r0.stdout.splitlines() if "clk" in l or "Power" in l), flush=True)
for name, N, d in (("headline 1e6 x 64", 1_000_000, 64), ("C5 share 5e5 x 256", 500_000, 256), ("2e6 x 32", 2_000_000, 32)):
    A = rng.standard_normal((d, d))
    D = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), A @ A.T / d + np.eye(d), ctx=ctx)
    X = torch.randn(N, d, dtype=torch.float64, device="cuda")
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    D.pdf_dev(X, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); D.pdf_dev(X, out); e1.record(); torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1) * 1e-3, 5e-5)
    n = int(4.0 / per)
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples)); th.start()
    t0 = time.perf_counter()
    e0.record()
    for i in range(n):
        D.pdf_dev(X, out)
        if i % 2000 == 1999:
            torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    mid = [s for s in samples if len(s) == 4 and s[0] - t0 > 1.0]
    print("%s: %.1f us per launch over %d launches; samples after the first second (sclk MHz, mclk MHz, W): %s" %
          (name, us, n, " ".join("%s/%s/%s" % (s[1][0] if s[1] else "?", s[2][0] if s[2] else "?", s[3][0] if s[3] else "?") for s in mid)), flush=True)
    if not mid:
        print("   raw: %r" % (samples[:3],))
    D.close(); del X, out
