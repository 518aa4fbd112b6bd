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


def _all_to_all(out, inp, out_splits, in_splits, group=None):
    """all_to_all_single; gloo has no device path for it, so a rehearsal on CUDA tensors (all ranks on one
    GPU over gloo) stages through the host -- the RCCL path never does."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group)
    return out


def exchange_row_table(x_local, a_local, n_total, group=None, stats=None, _always_exchange=False):
    """x_full[a_local[i]] for this rank's ancestors WITHOUT gathering x_full: every rank asks the owner
    of each ancestor for that row (one all-to-all of indices, 4 bytes each) and gets the rows back (one
    all-to-all of rows).  Per rank and step that is count * (R-1)/R rows in -- N/R of them -- where the
    all-gather of x_{t-1} brought in all N: 56 MB instead of 448 MB at N = 1e6, d = 64, R = 8
    (1.75 instead of 14 MB at d = 2).  x_local: this rank's rows [first, first+count) of x_{t-1};
    a_local: int32/int64 GLOBAL ancestor indices of this rank's particles.

    Returns (table, idx), idx int32: the ancestor row of particle i is table[idx[i]].  The rows are NOT
    put in particle order here: the proposal kernels gather through an index table anyway
    (cusmc_propagate_dev's a_dev), so handing them the received rows and the inverse permutation saves a
    pass over the rows; exchange_rows() materialises table[idx] for callers that want the rows.
    `stats`, if given, accumulates the bytes this rank sent and received."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1 and not _always_exchange:  # (_always_exchange: lets a one-rank RCCL group run the collectives, tests only)
        return x_local, a_local.to(torch.int32)
    counts = shard_counts(n_total, world)
    first = sum(counts[:rank])
    dev = x_local.device
    bounds = torch.tensor([sum(counts[:k + 1]) for k in range(world - 1)], dtype=torch.int64, device=dev)
    a64 = a_local.to(torch.int64)
    owner = torch.bucketize(a64, bounds, right=True)         # owner(j) = #{k : j >= first_{k+1}}
    order = torch.argsort(owner, stable=True)
    want = a64[order].to(torch.int32)                        # grouped by owner, 4 bytes per index
    send_n = torch.bincount(owner, minlength=world)
    recv_n = torch.empty_like(send_n)
    _all_to_all(recv_n, send_n, None, None, group)           # how many rows each peer asks of me
    send_l, recv_l = send_n.tolist(), recv_n.tolist()        # (host sync: the split sizes are data)
    asked = torch.empty(sum(recv_l), dtype=torch.int32, device=dev)
    _all_to_all(asked, want, recv_l, send_l, group)
    rows_out = x_local[(asked.to(torch.int64) - first)]      # the rows my peers asked for, in their order
    rows_in = torch.empty((sum(send_l),) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=dev)
    _all_to_all(rows_in, rows_out, send_l, recv_l, group)
    inv = torch.empty(order.shape[0], dtype=torch.int32, device=dev)
    inv[order] = torch.arange(order.shape[0], dtype=torch.int32, device=dev)  # particle order[k] asked for row k
    if stats is not None:
        row_b = x_local[0].numel() * x_local.element_size() if x_local.shape[0] else 0
        off = sum(send_l) - send_l[rank]
        stats["index_bytes_out"] = stats.get("index_bytes_out", 0) + 4 * off
        stats["row_bytes_in"] = stats.get("row_bytes_in", 0) + row_b * off
        stats["row_bytes_out"] = stats.get("row_bytes_out", 0) + row_b * (sum(recv_l) - recv_l[rank])
    return rows_in, inv


def exchange_rows(x_local, a_local, n_total, group=None, stats=None):
    """exchange_row_table() with the rows put in the order of a_local: count x d."""
    table, idx = exchange_row_table(x_local, a_local, n_total, group, stats)
    return table[idx.long()]


def run_filter_sharded(N, T, init_fn, step_fn=None, group=None, resample_fn=None, move_fn=None, stats=None):
    """The bootstrap filter's time loop (MCMC(), src/mcmc.cpp:292-308) with the particles sharded
    over the ranks -- the exact algorithm, not an island filter:

        x_0, w_0 = init_fn(first, count)                         this rank's rows of step 0
        for t = 1 .. T-1:
            w_full = all-gather(w_{t-1})                         8 N bytes: every chain reads w[j] anywhere
            a_t    = resample_fn(t, w_full, first, count)        this rank's chains (global indices)
            x_anc  = exchange_rows(x_{t-1}, a_t)                 only the N/R rows its ancestors name
            x_t, w_t = move_fn(t, x_anc, first, count)           propagate + reweight, local

    Everything is keyed by the GLOBAL particle index, so the concatenated shards equal the
    single-process run bit for bit.  Returns this rank's (X [T, count, d], w [T, count], a [T, count])
    -- ancestors of step 0 are zero, as in cusmc_pf_run_host.

    Legacy form (step_fn given): x_{t-1} is all-gathered whole every step (8 N d bytes) and
    step_fn(t, w_full, X_full, first, count) does the rest.  Kept for A/B timing of the two exchanges.

    The compute is passed in (gpu_filter_callables() builds it over the C ABI) so that the CPU tests can
    drive the same loop with stand-ins."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    first, count = shard_range(N, rank, world)
    x, w = init_fn(first, count)
    if w is None:  # initialize(): w_0 = 1/N (src/mcmc.cpp:85)
        w = torch.full((count,), 1.0 / N, dtype=torch.float64, device=x.device)
    X_hist, w_hist, a_hist = [x], [w], [torch.zeros(count, dtype=torch.int32, device=x.device)]
    for t in range(1, T):
        w_full = all_gather_ragged(w, N, group) if world > 1 else w
        if stats is not None and world > 1:
            stats["weight_bytes_in"] = stats.get("weight_bytes_in", 0) + 8 * (N - count)
        if step_fn is not None:
            X_full = all_gather_ragged(x, N, group) if world > 1 else x
            if stats is not None and world > 1:
                stats["row_bytes_in"] = stats.get("row_bytes_in", 0) + (N - count) * x[0].numel() * 8
            a, x, w = step_fn(t, w_full, X_full, first, count)
        else:
            a = resample_fn(t, w_full, first, count)
            if getattr(move_fn, "takes_row_table", False):  # (table, idx): the proposal kernel does the gather
                x_anc = exchange_row_table(x, a, N, group, stats) if world > 1 else (x, a.to(torch.int32))
            else:
                x_anc = exchange_rows(x, a, N, group, stats) if world > 1 else x[a.long()]
            x, w = move_fn(t, x_anc, first, count)
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


def gpu_filter_callables_exchange(Y, m0, C0, F, G, V, W, df=0.0, distribution="mvn", B=10, seed=0, compat=False,
                                  ctx=None):
    """init_fn / resample_fn / move_fn for run_filter_sharded's row-exchange form: resample ->
    (exchange) -> propagate -> reweight through the three separate C-ABI entry points, which give the
    same numbers as the fused step (tests/test_gpu_parity.py::test_fused_step_equals_three_launches) and
    hence as cusmc_pf_run_host.  Returns (init_fn, resample_fn, move_fn, obs)."""
    import numpy as np

    from . import api
    init_fn, _, obs = gpu_filter_callables(Y, m0, C0, F, G, V, W, df, distribution, B, seed, compat, ctx)
    ctx = obs.ctx
    Y = np.asarray(Y, dtype=np.float64)
    d = Y.shape[0]
    F, G, W = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (F, G, W))
    Qw = api.eigenSolver(W)
    scale = api.SQRT3 if compat else 1.0
    def resample_fn(t, w_full, first, count):
        a = torch.empty(count, dtype=torch.int32, device="cuda")
        api.Sampler.metropolis_hastings_dev(w_full, a, B=B, t=t, seed=seed, first=first, ctx=ctx)
        return a

    def move_fn(t, x_anc, first, count):
        # x_anc = (table, idx): particle i descends from table[idx[i]] -- the rows as they arrived from their
        # owners; the proposal kernel gathers through idx (cusmc_propagate_dev's a_dev), no reordering pass
        table, idx = x_anc
        x = torch.empty(count, d, dtype=torch.float64, device="cuda")
        w = torch.empty(count, dtype=torch.float64, device="cuda")
        api.propagate_dev(table.contiguous(), idx.contiguous(), G, Qw, x, kind=distribution, nu=df, scale=scale, seed=seed,
                          step=t, first=first, ctx=ctx)
        obs.reweight_dev(x, Y[:, t], F, w, log=False)
        return x, w
    move_fn.takes_row_table = True

    return init_fn, resample_fn, move_fn, obs
