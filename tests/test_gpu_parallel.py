"""The sharded pipeline END TO END on hardware: two processes share the box's one GPU, each runs the HIP
pack -> reduce on its own contiguous shard of sequences, and the [B/R, H] outputs are all-gathered.
RCCL refuses two ranks on one device, so the exchange rides on gloo here (RCCL itself is exercised by
bench.py under torchrun); what is under test is shard ownership, local PackedSequence construction, result
order, and that per-shard results equal the single-process result bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, H, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torchrua_amd as ta
    from torchrua_amd.parallel import all_gather_rows, shard_bounds
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        lens = torch.randint(1, 60, (B,), generator=g)
        data = torch.randn(int(lens.sum()), H, generator=g).to(torch.bfloat16)
        lo, hi = shard_bounds(B, rank, world)
        off = torch.cumsum(lens, 0) - lens
        rows = slice(int(off[lo]), int(off[hi - 1] + lens[hi - 1]))
        local = ta.with_host_sizes(data[rows].to(dev), lens[lo:hi])
        p = local.pack()                                   # this rank's own PackedSequence
        out_local = ta.reduce_sum(p)                       # [B/R, H] in local batch order
        gathered = all_gather_rows(out_local.cpu(), n_total=B).to(dev)
        whole = ta.reduce_sum(ta.with_host_sizes(data.to(dev), lens).pack())   # single-process result
        ok = torch.equal(gathered, whole) and gathered.shape == (B, H)
        ok = ok and int(p.batch_sizes[0]) == hi - lo
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('B', [512, 333])
def test_two_ranks_one_gpu(B):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == [(0, True), (1, True)]
