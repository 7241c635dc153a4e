"""Developer probe: the cfg3 / cfg2 reducers, many calls each (run under rocprofv3 --kernel-trace --stats to split
kernel time from launch gaps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
REP = int(os.environ.get('REP', 30))


def inputs(seed, B, lo, hi, H):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn(int(lens.sum()), H, device=dev).to(torch.bfloat16)
    return lens, data


def timed(name, fn, nbytes):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REP
    print(f'{name:28s} {ms * 1e3:8.1f} us/call back-to-back   {nbytes / ms / 1e9:5.2f} TB/s')


lens, data = inputs(3, 16384, 1, 64, 512)
ld = lens.to(dev)
nb = data.numel() * 2 + 16384 * 512 * 2
for name in ('sum', 'max', 'logsumexp', 'mean', 'min', 'prod'):
    fn = getattr(ta, f'segment_{name}')
    timed(f'cfg3 segment_{name}', lambda: fn(data, ld), nb)
lens2, data2 = inputs(2, 4096, 8, 512, 256)
c = ta.with_host_sizes(data2, lens2)
p = c.pack()
nb2 = data2.numel() * 2 + 4096 * 256 * 2
timed('cfg2 segment_sum(c)', lambda: ta.segment_sum(c.data, c.token_sizes), nb2)
timed('cfg2 reduce_sum(p)', lambda: ta.reduce_sum(p), nb2)
timed('cfg2 reduce_max(p)', lambda: ta.reduce_max(p), nb2)
