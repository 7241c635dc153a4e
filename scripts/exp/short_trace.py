"""Developer probe for rocprofv3 --kernel-trace --stats: reduce_sum / max / logsumexp over 1 M sequences of U(1,8) rows of
64 bytes (host-known lengths) and over 200 000 x U(1,32) rows of 16 bytes, 20 calls each — which kernel holds the gap
between max and sum at these shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
for B, lo, hi, H in ((1000000, 1, 8, 32), (200000, 1, 32, 8)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn(int(lens.sum()), H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    for z in (c, c.pack()):
        for name in ('sum', 'max', 'logsumexp'):
            for _ in range(20):
                getattr(ta, f'reduce_{name}')(z)
    torch.cuda.synchronize()
