"""Developer probe: kernel-only durations of the mid-size BASELINE configs (cfg2, cfg3) through the KernelTimer
hook (HIP events around each launch on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402

dev = torch.device('cuda:0')


class Timer:
    def __init__(self):
        self.open, self.ev = {}, []

    def __call__(self, name, begin):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if begin:
            self.open[name] = ev
        else:
            self.ev.append((name, self.open.pop(name), ev))

    def report(self, nbytes):
        torch.cuda.synchronize()
        by = {}
        for n, a, b in self.ev:
            by.setdefault(n, []).append(a.elapsed_time(b))
        for n, ts in by.items():
            ts.sort()
            ms = ts[len(ts) // 2]
            print(f'    kernel {n:12s} median {ms * 1e3:8.1f} us   {nbytes / ms / 1e9:5.2f} TB/s', flush=True)
        self.ev = []


def run(label, fn, nbytes, iters=20):
    t = Timer()
    fn()
    torch.cuda.synchronize()
    O.set_kernel_hook(t)
    for _ in range(iters):
        fn()
    O.set_kernel_hook(None)
    print(label)
    t.report(nbytes)


e = 2
for (B, lo, hi, H) in ((4096, 8, 512, 256), (16384, 1, 64, 512), (512, 8, 512, 512), (8192, 8, 512, 512)):
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = N * H * e
    print(f'--- B={B} U({lo},{hi}) H={H}: payload {nb / 1e6:.0f} MB')
    run('c.pack()', lambda: c.pack(), 2 * nb)
    run('reduce_sum(p)', lambda: ta.reduce_sum(p), nb)
    run('segment_sum(c)', lambda: ta.segment_sum(c.data, c.token_sizes), nb)
    run('segment_max(c)', lambda: ta.segment_max(c.data, c.token_sizes), nb)
    run('p.cat()', lambda: p.cat(), 2 * nb)

# split-policy sweep for the reducer (developer knob: rows per part, 0 = never split)
from torchrua_amd import _meta as M  # noqa: E402
orig = M.reduce_split_rows
print('=== reducer split sweep: segment_sum(c) / reduce_sum(p) kernel us')
for (B, lo, hi, H) in ((512, 8, 512, 512), (4096, 8, 512, 256), (8192, 8, 512, 512), (2048, 8, 512, 64), (16384, 8, 512, 512)):
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = N * H * e
    print(f'--- B={B} U({lo},{hi}) H={H}: payload {nb / 1e6:.0f} MB')
    for split in ('policy', 0, 32, 64, 128, 256):
        M.reduce_split_rows = orig if split == 'policy' else (lambda lay, rb=0, team_ok=True, s=split: s if lay.n_rows > s else 0)
        run(f'  split={split}: segment_sum(c)', lambda: ta.segment_sum(c.data, c.token_sizes), nb)
        run(f'  split={split}: reduce_sum(p)', lambda: ta.reduce_sum(p), nb)
    M.reduce_split_rows = orig
