"""Track sharding for multi-GPU batches: one process per GPU, contiguous track ranges,
no data-path collective.  torch.distributed is used for exactly three things: the barrier around the
timed region, the MAX-reduce of the step time, and handing per-track scalars (LUFS, dBTP, gain) to rank 0.
Works with backend "nccl" (RCCL, GPU tensors) and "gloo" (CPU tensors; tests/test_shard_gloo.py).
"""
import torch
import torch.distributed as dist


def shard_range(ntracks, rank, world):
    """Contiguous range [lo, hi) of rank `rank`: track i goes to rank floor(i * world / ntracks)."""
    if world < 1 or not (0 <= rank < world) or ntracks < 0:
        raise ValueError(f"bad shard request: ntracks={ntracks} rank={rank} world={world}")
    lo = -(-rank * ntracks // world)          # ceil(rank * ntracks / world): first i with floor(i * world / ntracks) == rank
    hi = -(-(rank + 1) * ntracks // world)
    return lo, hi


def owner_of(track, ntracks, world):
    return track * world // ntracks


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier(device=None):
    if _active():
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device="cpu"):
    """Step time of the slowest rank (the bench contract's MAX over ranks)."""
    if not _active():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_track_scalars(local, ntracks, device="cpu"):
    """`local` = float64 tensor [hi - lo, k] of per-track scalars of this rank's shard.
    Returns the [ntracks, k] table in track order on every rank (ranks own unequal counts when
    world does not divide ntracks, so shards are padded to the largest)."""
    local = torch.as_tensor(local, dtype=torch.float64, device=device)
    if local.dim() == 1:
        local = local[:, None]
    if not _active():
        assert local.shape[0] == ntracks
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    k = local.shape[1]
    most = max(shard_range(ntracks, r, world)[1] - shard_range(ntracks, r, world)[0] for r in range(world))
    pad = torch.full((most, k), float("nan"), dtype=torch.float64, device=device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.empty((ntracks, k), dtype=torch.float64, device=device)
    for r in range(world):
        lo, hi = shard_range(ntracks, r, world)
        out[lo:hi] = parts[r][: hi - lo]
    return out
