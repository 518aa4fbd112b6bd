"""CPU suite, part 3: the one-process-per-GPU partitioning logic, world_size 2 over gloo.
The compute is a stand-in (the oracle, allowed in tests/): what is under test is that the
sharded path produces exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_range_partitions():
    from cusmc_amd.sharding import shard_counts, shard_range
    for n in (0, 1, 7, 8, 1000, 10 ** 6 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(shard_counts(n, world)) - min(shard_counts(n, world)) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cusmc_amd.sharding import gather_final, shard_range, sharded_map, sharded_resample
    from oracle import oracle as O
    rng = np.random.default_rng(123)
    d = 8
    A = rng.standard_normal((d, d))
    sigma = A @ A.T / d + np.eye(d)
    X = rng.standard_normal((n, d))
    first, count = shard_range(n, rank, world)

    # leg 1: log-pdf of this rank's particles, no collective
    lp_local = sharded_map(torch.from_numpy(X[first:first + count]),
                           lambda x: torch.from_numpy(O.logpdf_hoisted(x.numpy(), None, sigma)))
    # leg 2: exact sharded resample = all-gather(w) + own index range, global Philox indices
    w_local = torch.exp(lp_local)

    def resample(w_full, f, c):
        full = O.metropolis(w_full.numpy(), 10, seed=77, step=2)  # stand-in computes all, keeps own
        return torch.from_numpy(full[f:f + c].astype(np.int64))

    a_local, w_full = sharded_resample(w_local, n, resample)
    a_full = gather_final(a_local, n)
    if rank == 0:
        np.savez(tmp, lp=w_full.numpy(), a=a_full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 101])
def test_two_rank_sharded_path_equals_single_process(tmp_path, n, oracle):
    tmp = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), n, tmp), nprocs=2, join=True)
    got = np.load(tmp)
    rng = np.random.default_rng(123)
    d = 8
    A = rng.standard_normal((d, d))
    sigma = A @ A.T / d + np.eye(d)
    X = rng.standard_normal((n, d))
    w = np.exp(oracle.logpdf_hoisted(X, None, sigma))
    assert np.array_equal(got["lp"], w)                      # ragged all-gather is exact
    assert np.array_equal(got["a"], oracle.metropolis(w, 10, seed=77, step=2))


def _filter_worker(rank, world, port, N, d, T, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cusmc_amd.sharding import gather_final, run_filter_sharded
    from oracle import oracle as O
    I = np.eye(d)
    rng = np.random.default_rng(5)
    Y = np.cumsum(0.1 * rng.standard_normal((T, d)), axis=0)
    G, Qw, Q0 = 0.9 * I, O.eigen_sqrt(0.1 * I), O.eigen_sqrt(I)
    V = 0.5 * I

    def init_fn(first, count):  # stand-in: the oracle draws all N rows, this rank keeps its own
        X0, _ = O.initialize(N, np.zeros(d), Q0, "mvn", 0.0, 1.0, seed=9, step=0)
        return torch.from_numpy(X0[first:first + count]), None

    def step_fn(t, w_full, X_full, first, count):
        a = O.metropolis(w_full.numpy(), 10, 9, step=t)
        X = O.propagate(X_full.numpy(), a, G, Qw, "mvn", 0.0, 1.0, seed=9, step=t)
        w = O.reweight(X, Y[t], I, V, "mvn", 0.0)
        sl = slice(first, first + count)
        return (torch.from_numpy(a[sl].astype(np.int32)), torch.from_numpy(X[sl].copy()), torch.from_numpy(w[sl].copy()))

    Xl, wl, al = run_filter_sharded(N, T, init_fn, step_fn)
    # the shards are rows [first, first+count) of every step: gather along the particle axis
    Xf = gather_final(Xl.permute(1, 0, 2).contiguous(), N).permute(1, 0, 2)
    wf = gather_final(wl.t().contiguous(), N).t()
    af = gather_final(al.t().contiguous(), N).t()
    if rank == 0:
        np.savez(tmp, X=Xf.numpy(), w=wf.numpy(), a=af.numpy(), Y=Y)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_filter_equals_single_process(tmp_path, oracle):
    """run_filter_sharded over gloo, world_size 2, with the oracle as the per-step compute: the
    gathered history equals the oracle's own single-process filter (same Philox keys, global
    particle indices)."""
    N, d, T = 203, 2, 6
    tmp = str(tmp_path / "pf.npz")
    mp.spawn(_filter_worker, args=(2, _free_port(), N, d, T, tmp), nprocs=2, join=True)
    got = np.load(tmp)
    I = np.eye(d)
    Xo, wo, ao = oracle.pf_run(got["Y"], N, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, "mvn", 0.0, B=10, seed=9)
    assert np.array_equal(got["a"][1:], ao[1:])
    assert np.allclose(got["X"], Xo, atol=1e-12) and np.allclose(got["w"], wo, rtol=1e-10)


def _exchange_worker(rank, world, port, N, d, T, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cusmc_amd.sharding import exchange_rows, gather_final, run_filter_sharded, shard_range
    from oracle import oracle as O
    first, count = shard_range(N, rank, world)

    # (1) exchange_rows alone: arbitrary (repeated, own and foreign) ancestors
    rng = np.random.default_rng(77)
    X_full = rng.standard_normal((N, d))
    a_full = rng.integers(0, N, size=N)
    a_full[::7] = a_full[0]  # heavy repeats, as after a resample
    stats = {}
    got = exchange_rows(torch.from_numpy(X_full[first:first + count].copy()),
                        torch.from_numpy(a_full[first:first + count].astype(np.int32)), N, stats=stats)
    assert np.array_equal(got.numpy(), X_full[a_full[first:first + count]])
    foreign = int(np.sum((a_full[first:first + count] < first) | (a_full[first:first + count] >= first + count)))
    assert stats["row_bytes_in"] == foreign * d * 8 and stats["index_bytes_out"] == foreign * 4

    # (2) the filter loop in its exchange form, the oracle as the per-step compute
    I = np.eye(d)
    rng = np.random.default_rng(5)
    Y = np.cumsum(0.1 * rng.standard_normal((T, d)), axis=0)
    G, Qw, Q0 = 0.9 * I, O.eigen_sqrt(0.1 * I), O.eigen_sqrt(I)
    V = 0.5 * I

    def init_fn(f, c):
        X0, _ = O.initialize(N, np.zeros(d), Q0, "mvn", 0.0, 1.0, seed=9, step=0)
        return torch.from_numpy(X0[f:f + c]), None

    def resample_fn(t, w_full, f, c):
        return torch.from_numpy(O.metropolis(w_full.numpy(), 10, 9, step=t)[f:f + c].astype(np.int32))

    def move_fn(t, x_anc, f, c):
        # the oracle keys its draws by the row index: place this rank's gathered rows at their global
        # positions and propagate with identity ancestors
        Xp = np.zeros((N, d))
        Xp[f:f + c] = x_anc.numpy()
        X = O.propagate(Xp, np.arange(N), G, Qw, "mvn", 0.0, 1.0, seed=9, step=t)[f:f + c].copy()
        return torch.from_numpy(X), torch.from_numpy(O.reweight(X, Y[t], I, V, "mvn", 0.0))

    fstats = {}
    Xl, wl, al = run_filter_sharded(N, T, init_fn, resample_fn=resample_fn, move_fn=move_fn, stats=fstats)
    assert fstats["weight_bytes_in"] == 8 * (N - count) * (T - 1)
    assert fstats["row_bytes_in"] <= count * d * 8 * (T - 1)  # never more than its own N/R rows per step
    Xf = gather_final(Xl.permute(1, 0, 2).contiguous(), N).permute(1, 0, 2)
    wf = gather_final(wl.t().contiguous(), N).t()
    af = gather_final(al.t().contiguous(), N).t()
    if rank == 0:
        np.savez(tmp, X=Xf.numpy(), w=wf.numpy(), a=af.numpy(), Y=Y)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_filter_row_exchange_equals_single_process(tmp_path, oracle, world):
    """The exchange form of run_filter_sharded (all-gather of w, all-to-all of only the ancestor rows) over
    gloo with 2 and 3 ranks (ragged shards): the gathered history equals the oracle's single-process filter."""
    N, d, T = 203, 2, 6
    tmp = str(tmp_path / "pfx.npz")
    mp.spawn(_exchange_worker, args=(world, _free_port(), N, d, T, tmp), nprocs=world, join=True)
    got = np.load(tmp)
    I = np.eye(d)
    Xo, wo, ao = oracle.pf_run(got["Y"], N, np.zeros(d), I, I, 0.9 * I, 0.5 * I, 0.1 * I, "mvn", 0.0, B=10, seed=9)
    assert np.array_equal(got["a"][1:], ao[1:])
    assert np.allclose(got["X"], Xo, atol=1e-12) and np.allclose(got["w"], wo, rtol=1e-10)
