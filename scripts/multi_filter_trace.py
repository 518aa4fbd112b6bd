"""One sharded filter run below the C ABI under CUSMC_TRACE=1 (and, under rocprofv3 --kernel-trace, the per-step
kernel durations and the gaps between a shard's consecutive steps).   python scripts/multi_filter_trace.py N d T ndev"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd
from cusmc_amd import _lib
L = _lib.lib(); ctx = cusmc_amd.api.default_context()
def ptr(a): return a.ctypes.data_as(C.c_void_p)
N, d, T, ndev = (int(v) for v in sys.argv[1:5])
rng = np.random.default_rng(1)
Y = np.ascontiguousarray(np.cumsum(0.1 * rng.standard_normal((T, d)), axis=0)); I = np.eye(d)
m0, C0, F, G, V, W = np.zeros(d), I.copy(), I.copy(), 0.95 * I, 0.5 * I, 0.1 * I
tail = (ptr(Y), N, d, T, ptr(m0), ptr(C0), ptr(F), ptr(G), ptr(V), ptr(W), C.c_float(0.0), b"metropolis", b"mvn", 10, C.c_double(1.0), 7, None, None, None)
os.environ["CUSMC_TRACE"] = "1"
for _ in range(2):
    t0 = time.perf_counter()
    if ndev == 0:
        rc = L.cusmc_pf_run_host(ctx._h, *tail)
    else:
        devs = (C.c_int * ndev)(*([0] * ndev))
        rc = L.cusmc_pf_run_multi_host(devs, ndev, *tail)
    assert rc == 0
    print(N, d, T, ndev, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
