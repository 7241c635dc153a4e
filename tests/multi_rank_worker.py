"""One rank of the multi-GPU parity check (started by tests/test_gpu_multi.py, one fresh process per GPU, BEFORE
anything in it touches a GPU): the batch is sharded contiguously over the ranks — ragged shards, B is not a multiple of
the world size — every rank packs and reduces ITS sequences on ITS card, and ONE RCCL all-gather over xGMI returns the
[B, H] result in global batch order on every rank.  Rank 0 compares it, bit for bit, with the single-process result of
the whole batch on its own card."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    B, H = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device('cuda', int(os.environ['LOCAL_RANK']))
    torch.cuda.set_device(dev)
    import torchrua_amd as ta
    from torchrua_amd.parallel import all_gather_rows, bind_rank_to_cpus, shard_bounds
    cpus = bind_rank_to_cpus(rank, world, dev.index)
    dist.init_process_group('nccl', device_id=dev)
    try:
        g = torch.Generator().manual_seed(23)
        lens = torch.randint(1, 200, (B,), generator=g)
        data = torch.randn(int(lens.sum()), H, generator=g).to(torch.bfloat16)
        lo, hi = shard_bounds(B, rank, world)
        off = torch.cumsum(lens, 0) - lens
        rows = slice(int(off[lo]), int(off[hi - 1] + lens[hi - 1]))
        ok = True
        for op in ('sum', 'max'):
            fn = ta.reduce_sum if op == 'sum' else ta.reduce_max
            p = ta.with_host_sizes(data[rows].to(dev), lens[lo:hi]).pack()          # this rank's own PackedSequence
            local = fn(p)                                                          # [B/R, H], local batch order
            gathered = all_gather_rows(local, n_total=B)                           # RCCL, ragged shards
            if B % world == 0:                                                     # equal shards: the async form too
                again, work = all_gather_rows(local, n_total=B, async_op=True)
                work.wait()
                ok = ok and torch.equal(again, gathered)
            torch.cuda.synchronize(dev)
            if rank == 0:
                whole = fn(ta.with_host_sizes(data.to(dev), lens).pack())
                ok = ok and gathered.shape == (B, H) and torch.equal(gathered, whole)
            ok = ok and int(p.batch_sizes[0]) == hi - lo
        flag = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(f'MULTI_RANK_OK world={world} B={B} cpus_rank0={len(cpus)}' if int(flag) == 1 else 'MULTI_RANK_MISMATCH', flush=True)
        code = 0 if int(flag) == 1 else 3
    finally:
        dist.destroy_process_group()
    sys.exit(code)


if __name__ == '__main__':
    main()
