"""Developer probe: does the north-star pack kernel's time depend on WHICH allocation it writes to / reads from?
bench.py alternates between two 17 GB output buffers and one of them is often 6 % slower (5.67 vs 6.00 ms)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
B, H = 65536, 512
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
nbytes = 2 * N * H * 2


def run(plan, src, out, reps=5):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        O.launch_move(plan, src, out=out)
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ts)[len(ts) // 2]


bufs = [torch.empty(N, H, device=dev, dtype=torch.bfloat16) for _ in range(int(os.environ.get('NBUF', 8)))]
bufs[0].normal_()
c = ta.with_host_sizes(bufs[0], lens)
p = c.pack()
del p
cl = describe(c)
pl = describe(ta.with_host_sizes(bufs[0], lens).pack())
plan = O.MovePlan(pl, cl, bufs[0].shape)
print('buffer addresses:', ' '.join(f'{b.data_ptr():#x}' for b in bufs))
print('C -> P, source = buffer 0, destination = buffer k')
for k in range(1, len(bufs)):
    ms = run(plan, bufs[0], bufs[k])
    print(f'  dst {k} at {bufs[k].data_ptr():#x} (delta {(bufs[k].data_ptr() - bufs[0].data_ptr()) / 2**30:+8.3f} GiB): {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s', flush=True)
print('C -> P, source = buffer k, destination = buffer 1 (source filled first)')
for k in range(2, len(bufs)):
    bufs[k].copy_(bufs[0])
    ms = run(plan, bufs[k], bufs[1])
    print(f'  src {k} at {bufs[k].data_ptr():#x}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s', flush=True)
print('pairs (src k, dst k+1)')
for k in range(0, len(bufs) - 1):
    if k:
        bufs[k].copy_(bufs[0])
    ms = run(plan, bufs[k], bufs[k + 1])
    print(f'  {k} -> {k + 1}: {ms:.3f} ms', flush=True)


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ts)[len(ts) // 2]


print('one direction only, per buffer: write (zero_), read (segment_sum over it as a CattedSequence)')
ld = c.token_sizes
for k in range(len(bufs)):
    w = timed(lambda: bufs[k].zero_())
    r = timed(lambda: ta.segment_sum(bufs[k], ld))
    print(f'  buffer {k}: write {w:.3f} ms ({nbytes / 2 / w / 1e9:.2f} TB/s)   read {r:.3f} ms ({nbytes / 2 / r / 1e9:.2f} TB/s)', flush=True)
print('plain copy_ between buffers (torch)')
for a, b in ((0, 1), (0, 2), (0, 7), (1, 2), (2, 3), (6, 7), (7, 0), (1, 0)):
    ms = timed(lambda: bufs[b].copy_(bufs[a]))
    print(f'  {a} -> {b}: {ms:.3f} ms', flush=True)
print('identity move through the mover (c.roll(0)-like C -> C)')
planc = O.MovePlan(cl, cl, bufs[0].shape)
for a, b in ((0, 1), (0, 2), (0, 7), (1, 2), (2, 3), (6, 7), (7, 0), (1, 0)):
    ms = run(planc, bufs[a], bufs[b])
    print(f'  {a} -> {b}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s', flush=True)
