"""Developer probe: rua_index_buckets against torch.sort(stable=True) at many (S, M), and its time at the north-star
size (17 M entries, 65 536 destinations) next to scatter_sum over the same entries."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator().manual_seed(0)
SIZES = () if os.environ.get('RUA_PROBE_TIMING_ONLY') else ((1, 1), (1, 10), (2, 64), (7, 5000), (255, 70000), (257, 70000), (511, 8192), (512, 8193), (65536, 300000),
             (70000, 123457), (3, 2049), (300000, 1 << 20), (65536, 8191), (5, 16385), (1 << 18, 3_000_001))
for S, M in SIZES:
    index = torch.randint(-2, S + 2, (M,), generator=g).to(dev)
    counts, perm = O.index_buckets(index, S)
    ok_idx = (index >= 0) & (index < S)
    key = torch.where(ok_idx, index, torch.full_like(index, S))
    order = torch.sort(key, stable=True)[1]
    n_ok = int(ok_idx.sum())
    assert torch.equal(perm[:n_ok], order[:n_ok]), (S, M)
    assert torch.equal(counts, torch.bincount(index[ok_idx], minlength=S)), (S, M)
    # skewed: nearly everything into one destination
    index = torch.zeros(M, dtype=torch.long)
    index[::7] = torch.randint(0, S, (index[::7].numel(),), generator=g)
    index = index.to(dev)
    counts, perm = O.index_buckets(index, S)
    assert torch.equal(perm, torch.sort(index, stable=True)[1]), ('skew', S, M)
print('bucketing == stable sort at every size', flush=True)

B, H = 65536, 512
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
index = torch.repeat_interleave(torch.arange(B), lens)
shuffled = index[torch.randperm(N, generator=g)].to(dev)
index = index.to(dev)
for name, ix in (('sorted index', index), ('shuffled index', shuffled)):
    ms = timeit(lambda: O.index_buckets(ix, B))
    print(f'index_buckets  {name:15s} {N} keys -> {B}: {ms * 1e3:8.1f} us', flush=True)
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
ten = torch.zeros(B, H, device=dev, dtype=torch.bfloat16)
for name, ix in (('sorted index', index), ('shuffled index', shuffled)):
    ms = timeit(lambda: ta.scatter_sum(ten, ix, data), iters=5)
    print(f'scatter_sum    {name:15s} {ms:8.3f} ms   ({(N * H * 2 + B * H * 2) / ms / 1e9:5.2f} TB/s algorithmic)', flush=True)
ms = timeit(lambda: ta.segment_sum(data, lens.to(dev)), iters=5)
print(f'segment_sum    {"":15s} {ms:8.3f} ms', flush=True)
