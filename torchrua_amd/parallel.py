"""Batch-sharded execution over several GPUs (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm, over xGMI).

Every op on the hot path is per-sequence, so rank r simply owns the contiguous shard of sequences
[r*B/R, (r+1)*B/R) and runs the single-GPU path on it: no payload ever crosses the fabric.  The only
exchange is ONE all-gather of the reduced [B/R, H] outputs; contiguous shards make the gathered
buffer come out in global batch order.  The reference has no distributed code."""
import os
from typing import Callable, List, Optional, Tuple

import torch.distributed as dist
from torch import Tensor

__all__ = ['shard_bounds', 'all_gather_rows', 'sharded_reduce', 'bind_rank_to_cpus']


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


# ------------------------------------------------------------------ host side of a multi-rank node
def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _pci_name(props) -> str:
    """sysfs name (dddd:bb:dd.f) of a GPU from torch's device properties: `pci_bus_id` is a string on some builds and
    the bus NUMBER (with `pci_domain_id` / `pci_device_id` beside it) on others."""
    bus = getattr(props, 'pci_bus_id', None)
    if isinstance(bus, str):
        return bus.lower()
    dom, dev = int(getattr(props, 'pci_domain_id', 0) or 0), int(getattr(props, 'pci_device_id', 0) or 0)
    return f'{dom:04x}:{int(bus):02x}:{dev:02x}.0'


def _numa_node_of_gpu(pci_name: str) -> int:
    """NUMA node of a GPU from sysfs (-1 when the platform does not say)."""
    for name in (pci_name, '0000:' + pci_name if pci_name.count(':') == 1 else pci_name):
        path = f'/sys/bus/pci/devices/{name}/numa_node'
        if os.path.exists(path):
            with open(path) as f:
                return int(f.read().strip())
    return -1


def plan_rank_cpus(local_rank: int, local_world: int, allowed: List[int], gpu_nodes: Optional[List[int]] = None,
                   node_cpus: Optional[dict] = None) -> List[int]:
    """The CPUs rank `local_rank` of `local_world` ranks on this host should run on: a share of the CPUs of ITS
    GPU's NUMA node (the pinned staging buffers, the host sort's scratch and the upload all live next to the card),
    split evenly among the ranks whose GPUs sit on that node; without topology information a contiguous 1/world slice
    of the allowed CPUs.  Pure function of its arguments (tests/test_parallel_gloo.py drives it with made-up
    topologies); never returns an empty list."""
    allowed = sorted(allowed)
    if local_world <= 1 or not allowed:
        return allowed
    if gpu_nodes and node_cpus and len(gpu_nodes) >= local_world and gpu_nodes[local_rank] in node_cpus:
        node = gpu_nodes[local_rank]
        mine = [c for c in node_cpus[node] if c in set(allowed)]
        peers = [r for r in range(local_world) if gpu_nodes[r] == node]
        if mine and len(mine) >= len(peers):
            k = peers.index(local_rank)
            share = len(mine) // len(peers)
            # SMT siblings are usually listed in the second half of a node's cpulist: deal the first half (one thread
            # per core) out first, then the siblings, so that every rank gets whole cores
            half = len(mine) // 2
            if half >= len(peers):
                # (a rank count that does not divide the cores leaves the remainder idle rather than splitting a
                # core's two threads between two ranks: six ranks on a 64-core node take 10 cores each)
                per = half // len(peers)
                return sorted(mine[k * per:(k + 1) * per] + mine[half + k * per:half + (k + 1) * per])
            return mine[k * share:(k + 1) * share]
    share = max(1, len(allowed) // local_world)
    return allowed[local_rank * share:(local_rank + 1) * share] or allowed


def bind_rank_to_cpus(local_rank: int, local_world: int, device_index: Optional[int] = None) -> List[int]:
    """Pin this process (one rank of a multi-GPU job) to its share of the host's CPUs and cap torch's intra-op
    thread pool to it; returns the CPU list (unchanged affinity when there is one rank, when RUA_NO_AFFINITY is set, or
    when the platform offers no sched_setaffinity).  Eight ranks on one host otherwise share every core: eight host
    sorts, eight 128-thread torch pools."""
    import torch
    if local_world <= 1 or os.environ.get('RUA_NO_AFFINITY') or not hasattr(os, 'sched_setaffinity'):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    allowed = sorted(os.sched_getaffinity(0))
    gpu_nodes, node_cpus = None, None
    try:
        gpu_nodes = [_numa_node_of_gpu(_pci_name(torch.cuda.get_device_properties(i)))
                     for i in range(torch.cuda.device_count())]
        if device_index is not None and device_index < len(gpu_nodes) and len(gpu_nodes) < local_world:
            # a rehearsal with every rank on ONE card (bench.py RUA_BENCH_DEVICE): all ranks sit on that card's node
            # and split its CPUs, which is what the ranks of one NUMA node of a multi-GPU host do
            gpu_nodes = [gpu_nodes[device_index]] * local_world
        elif device_index is not None and local_rank < len(gpu_nodes):
            gpu_nodes[local_rank] = gpu_nodes[device_index]
        node_cpus = {}
        base = '/sys/devices/system/node'
        for name in os.listdir(base):
            if name.startswith('node') and name[4:].isdigit():
                with open(os.path.join(base, name, 'cpulist')) as f:
                    node_cpus[int(name[4:])] = _parse_cpulist(f.read())
        if any(n < 0 for n in gpu_nodes[:local_world]):
            gpu_nodes = None
    except Exception:          # no sysfs, no pci_bus_id on this build: fall back to the contiguous split
        gpu_nodes, node_cpus = None, None
    cpus = plan_rank_cpus(local_rank, local_world, allowed, gpu_nodes, node_cpus)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return allowed
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
    return cpus
