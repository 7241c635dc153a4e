"""Developer A/B: independent waves per workgroup in seg_reduce_kernel at cfg3, cfg2 and the north-star shape.  The numbers in
profiles/r04_reduce_wpb_ab.txt came from a build with a temporary RUA_REDUCE_WPB=0|2|4 knob (the rule it found is now
fixed in launch_reduce); against HEAD the script times the shipped rule and labels the row with whatever the variable says.
(RUA_REDUCE_WPB=0|2|4) at cfg3, cfg2 and the
north-star shape; bursts of 8, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def burst(fn, reps=8, rounds=9):
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


tag = os.environ.get('RUA_REDUCE_WPB', '0')
for name, B, lo, hi, H in (('cfg3', 16384, 1, 64, 512), ('cfg2', 4096, 8, 512, 256), ('many-short', 262144, 1, 16, 512),
                           ('north-star', 65536, 8, 512, 512)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.empty(N, H, device=dev, dtype=torch.bfloat16).normal_()
    ld = lens.to(dev)
    c = ta.with_host_sizes(data, lens)
    nb = N * H * 2 + B * H * 2
    row = f'WPB={tag} {name:11s} N={N:9d}'
    for op in ('sum', 'max', 'logsumexp'):
        fn = getattr(ta, f'segment_{op}')
        us = burst(lambda: fn(data, ld), reps=8 if nb < 4e9 else 2)
        row += f' | segment_{op} {us:8.1f} us {nb / us / 1e6:5.2f} TB/s'
    if name != 'many-short':
        p = c.pack()
        us = burst(lambda: ta.reduce_sum(p), reps=8 if nb < 4e9 else 2)
        row += f' | reduce_sum(p) {us:8.1f} us {nb / us / 1e6:5.2f} TB/s'
        del p
    print(row, flush=True)
    del data, c
    torch.cuda.empty_cache()
