"""Batch-sharded execution over several GPUs (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm, over xGMI).

Every op on the hot path is per-sequence, so rank r simply owns the contiguous shard of sequences
[r*B/R, (r+1)*B/R) and runs the single-GPU path on it: no payload ever crosses the fabric.  The only
exchange is ONE all-gather of the reduced [B/R, H] outputs; contiguous shards make the gathered
buffer come out in global batch order.  The reference has no distributed code."""
from typing import Callable, Optional, Tuple

import torch.distributed as dist
from torch import Tensor

__all__ = ['shard_bounds', 'all_gather_rows', 'sharded_reduce']


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n sequences for `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError(f'rank {rank} outside world of {world}')
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: Tensor, n_total: Optional[int] = None, group=None, async_op: bool = False):
    """Concatenate every rank's [n_r, *H] rows in rank order.  Equal shards use one
    all_gather_into_tensor (a single ncclAllGather); shards that differ by a row are padded to the widest.
    async_op=True (equal shards only) returns (out, work): the collective runs on RCCL's own stream and
    overlaps whatever is enqueued next; call work.wait() before reading `out`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local = local.contiguous()
    if n_total is None:
        n_total = local.size(0) * world
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    if sizes[rank][1] - sizes[rank][0] != local.size(0):
        raise ValueError(f'rank {rank} holds {local.size(0)} rows, its shard of {n_total} is {sizes[rank]}')
    out = local.new_empty((n_total,) + tuple(local.shape[1:]))
    if n_total % world == 0:
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op:
        raise ValueError('async_op needs equal shards')
    # shards differ by one row: pad to the largest, gather once, drop the padding rows
    widest = max(hi - lo for lo, hi in sizes)
    padded = local.new_zeros((widest,) + tuple(local.shape[1:]))
    padded[:local.size(0)] = local
    tmp = local.new_empty((world * widest,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(tmp, padded, group=group)
    for r, (lo, hi) in enumerate(sizes):
        out[lo:hi] = tmp[r * widest:r * widest + (hi - lo)]
    return out


def sharded_reduce(local_fn: Callable[[], Tensor], n_total: Optional[int] = None, group=None) -> Tensor:
    """Run `local_fn` (this rank's pack -> reduce over its own sequences -> [B/R, H]) and all-gather."""
    return all_gather_rows(local_fn(), n_total=n_total, group=group)
