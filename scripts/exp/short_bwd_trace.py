"""Developer probe for rocprofv3 --kernel-trace --stats: the fused reduce backward over a PackedSequence and a CattedSequence of
SHORT sequences at narrow rows (200 000 x U(1,32) rows of 32 bytes fp32; 65 536 x U(1,16) rows of 256 bytes), 20 calls each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
for B, lo, hi, H in ((200000, 1, 32, 8), (65536, 1, 16, 64)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn(int(lens.sum()), H, device=dev, dtype=torch.float32)
    c = ta.with_host_sizes(data, lens)
    for z in (c, c.pack()):
        for name in ('sum', 'max', 'logsumexp'):
            x = z.data.detach().requires_grad_(True)
            out = getattr(ta, f'reduce_{name}')(z._replace(data=x))
            cot = torch.ones_like(out)
            for _ in range(20):
                torch.autograd.grad(out, x, cot, retain_graph=True)
    torch.cuda.synchronize()
