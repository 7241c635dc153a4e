"""Developer A/B: segment_sum over a CattedSequence with narrow rows — one wave per sequence against adjacent sequences
side by side in a wave (the PackedSequence reducer's RANKS form), by average length.  Run once per build
(RUA_LIB_PATH=scripts/exp/librua_catranks.so forces the side-by-side form)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def med(fn, rounds=7):
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


g = torch.Generator().manual_seed(1)
N0 = 8_000_000
for H in (8, 16, 32, 64, 128, 256):
    line = f'H={H:3d} ({H * 2:4d}-byte rows):'
    for avg in ([int(a) for a in os.environ["AVGS"].split(",")] if os.environ.get("AVGS") else (1, 2, 4, 8, 16, 32)):
        lens = torch.randint(1, 2 * avg, (N0 // avg,), generator=g) if avg > 1 else torch.ones(N0, dtype=torch.long)
        N = int(lens.sum())
        data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
        c = ta.with_host_sizes(data, lens)
        t = med(lambda: ta.reduce_sum(c))
        t2 = med(lambda: ta.reduce_max(c))
        line += f'  avg {avg:2d}: sum {t:6.3f} max {t2:6.3f} ms |'
        del data, c
    print(line, flush=True)
