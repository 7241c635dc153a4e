"""Developer probe: does arming the long-sequence split help mid-size CAT reductions (cfg2: 4 096 sequences, all
resident at once, the longest ones finish last)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, rep=40):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


orig = M.reduce_split_rows
for (B, lo, hi, H, tag) in ((4096, 8, 512, 256, 'cfg2'), (16384, 1, 64, 512, 'cfg3'), (2048, 8, 512, 512, 'B2048 H512'),
                            (8192, 8, 512, 512, 'B8192 H512'), (1024, 16, 1024, 1024, 'B1024 H1024')):
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn(int(lens.sum()), H, device=dev).to(torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = data.numel() * 2 + B * H * 2
    row = []
    for split in (None, 0, 32, 64, 128, 256):
        M.reduce_split_rows = orig if split is None else (lambda lay, rb=1024, team_ok=True, s=split: s)
        tc = timed(lambda: ta.segment_sum(c.data, c.token_sizes))
        tp = timed(lambda: ta.reduce_sum(p))
        tm = timed(lambda: ta.segment_max(c.data, c.token_sizes))
        row.append(f'split={split}: C sum {tc:6.1f} us ({nb / tc / 1e6:4.2f} TB/s)  C max {tm:6.1f}  P sum {tp:6.1f} ({nb / tp / 1e6:4.2f})')
    print(tag, f'{nb / 1e9:.2f} GB')
    for r in row:
        print('   ', r)
M.reduce_split_rows = orig
