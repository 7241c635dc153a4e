"""Developer probe: float scatter_sum under a skewed histogram when the host does not know the bucket sizes
(_meta.reduce_split_rows decides whether the long-sequence split is armed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')
Mn = 17_046_960


def med(fn, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


g = torch.Generator().manual_seed(3)
for H in (64, 512):
    src = torch.randn(Mn, H, dtype=torch.bfloat16, device=dev)
    for S in (65536, 100000, 1 << 18):
        for name, idx in (('uniform', torch.randint(0, S, (Mn,), generator=g)),
                          ('a third in one bucket', torch.where(torch.rand(Mn, generator=g) < 0.33, torch.tensor(5), torch.randint(0, S, (Mn,), generator=g))),
                          ('3 % in one bucket', torch.where(torch.rand(Mn, generator=g) < 0.03, torch.tensor(5), torch.randint(0, S, (Mn,), generator=g)))):
            idx = idx.to(dev)
            ten = torch.zeros(S, H, dtype=torch.bfloat16, device=dev)
            t = med(lambda: ta.scatter_sum(ten, idx, src))
            print(f'H={H:3d} S={S:7d} {name:24s}: scatter_sum {t:9.3f} ms', flush=True)
