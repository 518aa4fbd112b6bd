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
