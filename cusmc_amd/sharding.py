"""One process per GPU: how particles / chains are partitioned over the ranks of one node.

The reference has no multi-device code at all (zero collective call
sites), so this layer is new.  It follows what the hot path allows:

  * log-pdf / reweight: particles [first, first+count) live on rank r; parameters are a few KB
    and every rank factors Sigma itself.  NO collective in the data path.
  * Metropolis resampler: chain i only writes a[i], but reads w[j] for arbitrary j, so every
    rank needs the FULL weight vector: one all-gather of the weight shards (8 MB at N = 1e6;
    over xGMI each rank sends its 1/R shard to R-1 peers on direct links), then each rank
    resamples its own range with the shared Philox contract -- indices are identical to the
    single-GPU run because draws are keyed by the GLOBAL chain index.
  * final posterior draws / ancestors: one all-gather, only if the caller wants them whole.

torch.distributed is plumbing here (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).
The compute is passed in as a callable so that the CPU tests can drive this logic without a GPU.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous balanced split of range(n): returns (first, count) of `rank`."""
    base, extra = divmod(int(n), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def shard_counts(n, world):
    return [shard_range(n, r, world)[1] for r in range(world)]


def all_gather_ragged(local, n_total, group=None):
    """Concatenate the ranks' 1-D / row-major shards (possibly of unequal length) into the full
    array on every rank.  Equal shards use one all_gather_into_tensor; ragged ones pad to the
    longest shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_total, world)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    tail = tuple(local.shape[1:])
    if len(set(counts)) == 1 and dist.get_backend(group) != "gloo":
        full = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        return full
    longest = max(counts)
    padded = torch.zeros((longest,) + tail, dtype=local.dtype, device=local.device)
    padded[: counts[rank]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def sharded_resample(w_local, n_total, resample_fn, group=None):
    """Exact (not island) sharded Metropolis resampling.

    w_local      this rank's shard of the weight vector
    resample_fn  (w_full, first, count) -> ancestors of chains [first, first+count)
    returns      (a_local, w_full)
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    w_full = all_gather_ragged(w_local, n_total, group) if world > 1 else w_local
    first, count = shard_range(n_total, rank, world)
    return resample_fn(w_full, first, count), w_full


def sharded_map(x_local, fn):
    """Embarrassingly parallel leg (log-pdf, reweight): no communication at all."""
    return fn(x_local)


def gather_final(a_local, n_total, group=None):
    """The one collective of the sharded path's output side: the final draws / ancestors."""
    if dist.get_world_size(group) == 1:
        return a_local
    return all_gather_ragged(a_local, n_total, group)


def run_filter_sharded(N, T, init_fn, step_fn, group=None):
    """The bootstrap filter's time loop (MCMC(), src/mcmc.cpp:292-308) with the particles sharded
    over the ranks -- the exact algorithm, not an island filter:

        x_0, w_0 = init_fn(first, count)                         this rank's rows of step 0
        for t = 1 .. T-1:
            w_full = all-gather(w_{t-1});  X_full = all-gather(x_{t-1})
            a_t, x_t, w_t = step_fn(t, w_full, X_full, first, count)   this rank's rows of step t

    Chain i reads w_{t-1}[j] for arbitrary j and particle i reads x_{t-1}[a_i] from anywhere, so
    both vectors are gathered whole once per step (8 N and 8 N d bytes: 8 + 16 MB at N = 1e6,
    d = 2); everything else is local and keyed by the GLOBAL particle index, so the concatenated
    shards equal the single-process run.  Returns this rank's (X [T, count, d], w [T, count],
    a [T, count]) -- ancestors of step 0 are zero, as in cusmc_pf_run_host.

    The compute is passed in (gpu_filter_callables() builds it over cusmc_pf_step_dev) so that the
    CPU tests can drive the same loop with stand-ins."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    first, count = shard_range(N, rank, world)
    x, w = init_fn(first, count)
    if w is None:  # initialize(): w_0 = 1/N (src/mcmc.cpp:85)
        w = torch.full((count,), 1.0 / N, dtype=torch.float64, device=x.device)
    X_hist, w_hist, a_hist = [x], [w], [torch.zeros(count, dtype=torch.int32, device=x.device)]
    for t in range(1, T):
        w_full = all_gather_ragged(w, N, group) if world > 1 else w
        X_full = all_gather_ragged(x, N, group) if world > 1 else x
        a, x, w = step_fn(t, w_full, X_full, first, count)
        X_hist.append(x)
        w_hist.append(w)
        a_hist.append(a)
    return torch.stack(X_hist), torch.stack(w_hist), torch.stack(a_hist)


def gpu_filter_callables(Y, m0, C0, F, G, V, W, df=0.0, distribution="mvn", B=10, seed=0, compat=False, ctx=None):
    """init_fn / step_fn for run_filter_sharded over the HIP library on this rank's GPU: the same
    kernels, parameters and Philox keys as cusmc_pf_run_host (cusmc_amd.run), restricted to the rows
    [first, first + count).  Y is d x T (columns = time), as run() takes it."""
    import numpy as np

    from . import api
    ctx = ctx or api.default_context()
    ctx.use_torch_stream()
    Y = np.asarray(Y, dtype=np.float64)
    d = Y.shape[0]
    m0, C0, F, G, V, W = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (m0, C0, F, G, V, W))
    Q0, Qw = api.eigenSolver(C0), api.eigenSolver(W)
    scale = api.SQRT3 if compat else 1.0
    obs = (api.MultiVariateNormalDistribution(None, V, ctx=ctx) if distribution == "mvn"
           else api.MultiVariateTStudentDistribution(None, V, df, ctx=ctx))
    def init_fn(first, count):
        x = torch.empty(count, d, dtype=torch.float64, device="cuda")
        api.initialize_dev(m0, Q0, x, distribution, df, scale, seed=seed, first=first, ctx=ctx)
        return x, None  # w_0 = 1/N: filled in by run_filter_sharded

    def step_fn(t, w_full, X_full, first, count):
        a = torch.empty(count, dtype=torch.int32, device="cuda")
        x = torch.empty(count, d, dtype=torch.float64, device="cuda")
        w = torch.empty(count, dtype=torch.float64, device="cuda")
        api.pf_step_dev(obs, w_full, X_full, G, Qw, Y[:, t], F, a, x, w, kind=distribution, nu=df, B=B,
                        scale=scale, seed=seed, step=t, first=first)
        return a, x, w

    return init_fn, step_fn, obs
