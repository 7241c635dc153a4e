"""Developer probe: wall time of the host-side pieces of C.pack() + reduce_sum at B = 4096 (no profiler in the way)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as K, _meta as M, _ops as O  # noqa: E402
from torchrua_amd.core import _pack_meta, _hidden  # noqa: E402
from torchrua_amd.layout import P, describe  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
lens = torch.randint(1, 5, (4096,), generator=g)
data = torch.randn(int(lens.sum()), 256, device=dev, dtype=torch.bfloat16)
acc = {}


def lap(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def full():
    t = time.perf_counter()
    c = ta.with_host_sizes(data, lens)
    t = lap('with_host_sizes', t)
    lens_d, sorted_indices, unsorted, batch_sizes, bsz_dev, boff = _pack_meta(c.token_sizes, dev)
    t = lap('_pack_meta', t)
    n = int(c.data.size(0))
    shell = P(data=c.data, batch_sizes=batch_sizes, sorted_indices=sorted_indices, unsorted_indices=unsorted)
    M.adopt_pack(shell, lens_d, boff, bsz_dev)
    t = lap('P() + adopt_pack', t)
    dst = M.lay_pack(shell, lens=lens_d, boff=boff, T=batch_sizes.numel(), n_rows=n, row_bytes=M.row_bytes(c.data, 1))
    src = describe(c)
    t = lap('lay_pack + describe', t)
    plan = O.MovePlan(dst, src, (n,) + _hidden(c), name='to_pack')
    t = lap('MovePlan', t)
    d = O.move(c.data, plan)
    t = lap('O.move', t)
    p = shell._replace(data=d)
    t = lap('_replace (+ a ~19 us stall that follows the launch by ~2 us, whatever runs then)', t)
    out = ta.reduce_sum(p)
    t = lap('reduce_sum', t)
    return out


for _ in range(200):
    full()
torch.cuda.synchronize()
acc.clear()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    full()
total = (time.perf_counter() - t0) / N * 1e6
torch.cuda.synchronize()
for k, v in acc.items():
    print(f'{k:24s} {v / N * 1e6:7.1f} us')
print(f'{"total":24s} {total:7.1f} us')
