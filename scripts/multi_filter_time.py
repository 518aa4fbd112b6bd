"""The sharded filter loop below the C ABI against the one-device loop, outputs NOT copied out (NULL pointers: what
is timed is set-up + the T - 1 steps + teardown), BASELINE configs[2] (N = 1e6, d = 2, T = 100) and a d = 64 case.
On a one-GPU box the shards share device 0 (the rehearsal VERDICT r02 item 3 asks for): the same code path as
distinct devices except for the peer-copy branches.   python scripts/multi_filter_time.py [tag]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cusmc_amd  # noqa: E402
from cusmc_amd import _lib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
L = _lib.lib()
ctx = cusmc_amd.api.default_context()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def run(N, d, T, devices, reps=3):
    rng = np.random.default_rng(1)
    Y = np.ascontiguousarray(np.cumsum(0.1 * rng.standard_normal((T, d)), axis=0))
    I = np.eye(d)
    m0, C0, F, G, V, W = np.zeros(d), I.copy(), I.copy(), 0.95 * I, 0.5 * I, 0.1 * I
    tail = (ptr(Y), N, d, T, ptr(m0), ptr(C0), ptr(F), ptr(G), ptr(V), ptr(W), C.c_float(0.0), b"metropolis", b"mvn", 10,
            C.c_double(1.0), 7, None, None, None)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        if devices is None:
            rc = L.cusmc_pf_run_host(ctx._h, *tail)
        else:
            devs = (C.c_int * len(devices))(*devices)
            rc = L.cusmc_pf_run_multi_host(devs, len(devices), *tail)
        assert rc == 0, L.cusmc_last_error()
        best = min(best, time.perf_counter() - t0)
    return best


for N, d, T in ((1_000_000, 2, 100), (1_000_000, 2, 400), (200_000, 64, 20), (1000, 2, 1000)):
    one = run(N, d, T, None)
    print("%s N=%d d=%d T=%d one device: %.2f ms (%.1f us per step)" % (tag, N, d, T, one * 1e3, one * 1e6 / (T - 1)), flush=True)
    for devs in ([0, 0], [0, 0, 0, 0]):
        t = run(N, d, T, devs)
        print("%s N=%d d=%d T=%d devices=%s: %.2f ms (%.1f us per step, %.2f x one device)" % (
            tag, N, d, T, devs, t * 1e3, t * 1e6 / (T - 1), t / one), flush=True)
