"""Developer probe: what arming the reducer's long-sequence split costs when no sequence is long (device-only lengths),
by payload size: segment_sum over a CattedSequence whose lengths live on the device, bf16, H = 64 and 512."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')


def med(fn, reps=20, rounds=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2] * 1e3


orig = M.reduce_split_rows
g = torch.Generator().manual_seed(1)
for H in (64, 512):
    for B, hi in ((2048, 64), (16384, 64), (65536, 64), (16384, 200), (65536, 200)):
        lens = torch.randint(1, hi + 1, (B,), generator=g)
        N = int(lens.sum())
        data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
        dl = lens.to(dev)
        t = {}
        for arm in (False, True):
            def rule(lay, row_bytes=1024, team_ok=True, tail_ok=True, _arm=arm):
                if lay.max_len is not None:
                    return orig(lay, row_bytes, team_ok, tail_ok)
                if not _arm:
                    return 0
                lay.max_len = None
                n = lay.n_rows
                rb = max(1, min(int(row_bytes), 1024))
                part = max(32, (64 << 10) // rb, min(4096, n // 8192))
                ideal = int(0.75 * n * row_bytes / 5e12 * 4e9 / rb) + int(30e-6 * 4e9 / rb)
                part = max(part, min(4096, ideal))
                return part if n > part else 0
            M.reduce_split_rows = rule
            t[arm] = med(lambda: ta.segment_sum(data, dl))
        M.reduce_split_rows = orig
        print(f'H={H:3d} B={B:6d} U(1,{hi:3d}) payload {N * H * 2 / 1e6:8.1f} MB: unarmed {t[False]:7.1f} us, armed {t[True]:7.1f} us ({(t[True] / t[False] - 1) * 100:+5.1f} %)', flush=True)
