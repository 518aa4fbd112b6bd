"""CPU suite: bench.py's launcher contract (`--gpus N` must really start N ranks)."""
import importlib.util
import os
import subprocess
import sys
import types

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_without_launcher_spawns_that_many_ranks(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--no-pmc"])
    assert bench.main() == 7  # the children's exit code is passed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--no-pmc"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--no-pmc", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
