"""The collectives of cusmc_amd/sharding.py and bench.py over the REAL backend of the multi-GPU runs -- RCCL
(torch.distributed "nccl") -- as far as a one-GPU box allows: a one-rank process group.  Two ranks cannot share
a device under RCCL, so the world-2 and world-3 LOGIC is covered over gloo (tests/test_sharding.py, and on the GPU
in test_gpu_parity.py::test_sharded_filter_on_gpu_equals_run); what gloo cannot show is whether RCCL accepts the
calls as they are made: device tensors, float64 / int64 / int32 payloads, all_to_all_single with explicit split
lists, all_gather_into_tensor, MAX all-reduce, barrier with a device id.  A child process, so that the group's
lifetime is its own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, socket
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from cusmc_amd import sharding
dist.barrier()
t = torch.tensor([3.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
g = torch.Generator(device="cuda").manual_seed(5)
N, d = 10_007, 6
x = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
w = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
a = torch.randint(0, N, (N,), dtype=torch.int32, device="cuda", generator=g)
assert torch.equal(sharding.all_gather_ragged(w, N), w)            # all_gather_into_tensor, float64
assert torch.equal(sharding.all_gather_ragged(x, N), x)            # ... of rows
assert torch.equal(sharding.gather_final(a, N), a)
counts = torch.tensor([N], dtype=torch.int64, device="cuda")
got = torch.empty_like(counts)
sharding._all_to_all(got, counts, None, None)                      # int64, equal split
assert torch.equal(got, counts)
idx = torch.empty(N, dtype=torch.int32, device="cuda")
sharding._all_to_all(idx, a, [N], [N])                             # int32, explicit split lists
assert torch.equal(idx, a)
rows = torch.empty_like(x)
sharding._all_to_all(rows, x, [N], [N])                            # float64 rows, explicit split lists
assert torch.equal(rows, x)
st = {}
table, inv = sharding.exchange_row_table(x, a, N, stats=st, _always_exchange=True)   # the whole exchange, collectives included
assert torch.equal(table[inv.long()], x[a.long()])
a_res, w_full = sharding.sharded_resample(w, N, lambda wf, first, count: torch.arange(first, first + count, device="cuda"))
assert torch.equal(w_full, w) and a_res.shape[0] == N
dist.barrier()
dist.destroy_process_group()
print("RCCL one-rank collectives OK")
"""


def test_sharding_collectives_over_rccl():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL one-rank collectives OK" in out.stdout


def test_bench_under_the_launcher_with_an_rccl_group():
    """bench.py exactly as the driver starts it for N > 1 -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...` -- with N = 1 and
    CUSMC_BENCH_FORCE_DIST=1, so that the process group is RCCL and every barrier and MAX reduction of the timed
    region and of the legs goes through it.  One JSON line, the contract's keys, backend nccl."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUSMC_BENCH_FORCE_DIST="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--no-pmc", "--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["backend"] == "nccl" and line["ranks"] == 1 and line["n_gpus"] == 1
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["steps"] == 20 and line["warmup"] == 5 and line["dtype"] == "f64"
    assert 0.3 < line["roofline"]["frac"] < 1.0 and line["roofline"]["bound"] == "hbm"
    assert "aux_error" not in line
    for leg in ("mh", "strong", "filter_step"):
        assert line[leg] and "error" not in line[leg], (leg, line[leg])
