"""Developer probe: the two cliffs that only showed with DEVICE-ONLY lengths (the reference's signature
`segment_*(tensor, sizes_on_device)`), each against the same call with a host mirror of the lengths (with_host_sizes):
  (a) max / min / logsumexp over a batch that is mostly EMPTY sequences (the empty rows take the global extreme);
  (b) reductions at rows of <= 32 bytes with a payload / an average length large enough to arm the long-sequence split
      for lengths nobody vouches for (rounds 1-4: that sent them back to one wave per sequence).
bf16, HIP events, median of 7."""
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
print('== (a) 150 000 sequences, nine in ten empty, the others U(8,512) rows: ms per call, host-known / device-only lengths')
lens = torch.where(torch.rand(150_000, generator=g) < 0.9, torch.tensor(0), torch.randint(8, 513, (150_000,), generator=g))
N = int(lens.sum())
for H in (512, 64, 16):
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    host, devl = ta.with_host_sizes(data, lens), ta.C(data, lens.to(dev))
    line = f'H={H:4d} ({H * 2:5d}-byte rows, {N * H * 2 / 1e9:5.2f} GB, {int((lens == 0).sum()) * H * 2 / 1e6:6.1f} MB of empty rows) |'
    for name in ('sum', 'max', 'logsumexp'):
        fn = getattr(ta, f'reduce_{name}')
        line += f' {name} {med(lambda: fn(host)):7.3f} / {med(lambda: fn(devl)):7.3f} |'
    hp = host.pack()
    line += f' max(P) {med(lambda: ta.reduce_max(hp)):7.3f} |'
    print(line, flush=True)
    del data, host, devl, hp

print('== (b) narrow rows, lengths U(8,512) (+ one 2 M-row sequence in the second line of each width): ms and TB/s of payload, host-known / device-only')
for H in (8, 16):
    for giant in (False, True):
        rows = int(1.2e9 / (H * 2))
        B = rows // 260
        lens = torch.randint(8, 513, (B,), generator=g)
        if giant:
            lens[B // 3] = 2_000_000
        N = int(lens.sum())
        data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
        host, devl = ta.with_host_sizes(data, lens), ta.C(data, lens.to(dev))
        nb = N * H * 2
        line = f'H={H:3d} ({H * 2:3d}-byte rows) B={B:8d} N={N:10d} giant={int(giant)} |'
        for name in ('sum', 'max', 'logsumexp'):
            fn = getattr(ta, f'reduce_{name}')
            th, td = med(lambda: fn(host)), med(lambda: fn(devl))
            line += f' {name} {th:6.3f} ({nb / th / 1e9:4.2f}) / {td:6.3f} ({nb / td / 1e9:4.2f}) |'
        print(line, flush=True)
        del data, host, devl
        torch.cuda.empty_cache()
