"""The N>1 path on CPU: world_size 2 over gloo.  The per-rank compute is the HIP path on a GPU box; here
(no GPU) the oracle stands in for it so that sharding + the all-gather are what is under test."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import orc
from torchrua_amd.parallel import all_gather_rows, plan_rank_cpus, shard_bounds, sharded_reduce


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 8, 65536, 524288, 524291):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_rank_cpu_plan():
    """Every rank of a node gets its own cores, next to its card, whole cores first; no topology -> an even split."""
    nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    gpus = [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [plan_rank_cpus(r, 8, list(range(256)), gpus, nodes) for r in range(8)]
    assert all(len(p) == 32 for p in plans) and len({c for p in plans for c in p}) == 256
    assert plans[0][:16] == list(range(16)) and plans[0][16:] == list(range(128, 144))     # cores + their siblings
    assert all(c in nodes[1] for c in plans[5])
    # a cgroup that allows only 16 CPUs, no topology: contiguous quarters
    assert [plan_rank_cpus(r, 4, list(range(16))) for r in range(4)] == [list(range(4 * r, 4 * r + 4)) for r in range(4)]
    assert plan_rank_cpus(0, 1, [3, 4]) == [3, 4]
    assert plan_rank_cpus(2, 4, [7]) == [7]                 # fewer CPUs than ranks: never an empty set
    # GPUs 0-1 of a two-rank job on one node of a larger machine
    assert plan_rank_cpus(1, 2, list(range(256)), gpus, nodes) == list(range(32, 64)) + list(range(160, 192))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, H, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        lens = torch.randint(1, 20, (B,), generator=g)
        data = torch.randn(int(lens.sum()), H, generator=g)
        lo, hi = shard_bounds(B, rank, world)
        off = torch.cumsum(lens, 0) - lens
        row_lo = int(off[lo]) if lo < B else int(lens.sum())
        row_hi = int(off[hi - 1] + lens[hi - 1]) if hi > lo else row_lo

        def local():   # this rank's pack -> reduce over its own sequences only
            l, d = lens[lo:hi].numpy(), data[row_lo:row_hi].numpy()
            p = orc.to_pack(orc.C(d, l), orc.stable_descending_order(l))
            return torch.from_numpy(orc.segment_sum(orc.to_cat(p).data, l))

        out = sharded_reduce(local, n_total=B)
        full = torch.from_numpy(orc.segment_sum(data.numpy(), lens.numpy()))
        ok = torch.equal(out, full) and out.shape == (B, H)
        same = all_gather_rows(torch.full((1, 2), float(rank)))
        ok = ok and same[:, 0].tolist() == [float(r) for r in range(world)]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run(B, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == [(r, True) for r in range(world)]


def test_two_rank_gather_even():
    _run(64)       # equal shards: one all_gather_into_tensor


def test_two_rank_gather_ragged():
    _run(37)       # 19 + 18 sequences: the ragged all_gather path


def test_eight_rank_gather_even():
    """cfg5's rank count (8 x contiguous batch shards, one all-gather of the reduced rows), rehearsed over gloo."""
    _run(64 * 8, world=8)


def test_eight_rank_gather_ragged():
    _run(509 * 8 + 3, world=8)       # shards of 510 / 509 sequences: the ragged all-gather with eight ranks


def test_bench_refuses_or_accepts_eight_ranks_as_self_launch_documents(tmp_path):
    """`python bench.py --gpus 8` on a machine that shows fewer than eight GPUs must exit non-zero naming the shortfall
    (self_launch: never a line for another rank count); with RUA_BENCH_DEVICE set — every rank pinned to one card, the
    rehearsal mode — the check is waived and the ranks are started (here there is no card at all, so they fail at their
    first GPU call: what matters is that the launcher got as far as starting them and reports THEIR failure)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RUA_BENCH_DEVICE')}
    if torch.cuda.device_count() < 8:
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                             capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert out.returncode != 0 and '--gpus 8' in out.stderr and 'GPU' in out.stderr
        assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    if torch.cuda.device_count() == 0:
        env.update(RUA_BENCH_DEVICE='0', RUA_BENCH_BACKEND='gloo')
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0',
                              '--batch', '64', '--hidden', '8'], capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert out.returncode != 0 and 'exited with' in out.stderr          # the children were started and failed by themselves
        assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
