"""Developer probe: the reducer on a skewed batch (one very long sequence among short ones), with and
without the long-sequence splitting."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')
H = 512
lens = torch.full((2048,), 64, dtype=torch.long)
lens[777] = 1_000_000
data = torch.randn(int(lens.sum()), H, device=dev, dtype=torch.bfloat16)
c = ta.with_host_sizes(data, lens)
p = c.pack()
nbytes = data.numel() * 2


def t(name, fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f'{name:34s} {ms:9.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s')


policy = M.reduce_split_rows
for split in ('policy', 64, 256, 4096, 0):
    M.reduce_split_rows = policy if split == 'policy' else (lambda lay, rb=0, team_ok=True, s=split: s)
    tag = f'split {split}' if split else 'no split  '
    if split == 'policy':
        tag += f' (= {policy(M.lay_cat(c.token_sizes, lens.numel(), data.size(0)), H * 2)})'
    t(f'segment_sum(C)  {tag}', lambda: ta.segment_sum(c.data, c.token_sizes))
    t(f'reduce_sum(P)   {tag}', lambda: ta.reduce_sum(p))
    t(f'reduce_max(P)   {tag}', lambda: ta.reduce_max(p))
    x = c.data.detach().requires_grad_(True)
    out = ta.segment_max(x, c.token_sizes)
    t(f'segment_max backward  {tag}', lambda: torch.autograd.grad(out, x, torch.ones_like(out), retain_graph=True))
a = ta.reduce_sum(p)
M.reduce_split_rows = policy
b = ta.reduce_sum(p)
print('max |split - nosplit| =', (a.float() - b.float()).abs().max().item())
