"""Developer probe: why the reduce over a PackedSequence beats the reduce over a CattedSequence at cfg3 (16 384 segments
U(1,64), 1-KiB rows) by ~10 %: the same segments in random order, sorted longest first (what a PackedSequence's rank order
is), and shortest first; bursts of 8, us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def burst(fn, reps=8, rounds=15):
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


g = torch.Generator().manual_seed(3)
lens = torch.randint(1, 65, (16384,), generator=g)
data = torch.randn(int(lens.sum()), 512, device=dev, dtype=torch.bfloat16)
for tag, l in (('random order', lens), ('longest first', lens.sort(descending=True)[0]), ('shortest first', lens.sort()[0]),
               ('all equal (32)', torch.full((int(lens.sum()) // 32,), 32))):
    n = int(l.sum())
    c = ta.C(data[:n], l.to(dev))
    h = ta.with_host_sizes(data[:n], l)
    print(f'{tag:16s} C device {burst(lambda: ta.reduce_sum(c)):7.1f}  C host {burst(lambda: ta.reduce_sum(h)):7.1f}  '
          f'P {burst(lambda: ta.reduce_sum(h.pack())):7.1f} (incl. pack)  max(C device) {burst(lambda: ta.reduce_max(c)):7.1f}', flush=True)
p = ta.with_host_sizes(data, lens).pack()
print(f'P of the random order: sum {burst(lambda: ta.reduce_sum(p)):7.1f}  max {burst(lambda: ta.reduce_max(p)):7.1f}')
