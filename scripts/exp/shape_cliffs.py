"""Developer probe: the movers and reducers on degenerate length distributions against a uniform one of the same size
(bf16, H = 512 and H = 16): one giant sequence, mostly-empty batches, all sequences of length 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
if os.environ.get('RUA_PROBE_NO_TILES'):      # narrow rows through the generic mover (no tile table handed over)
    from torchrua_amd import _meta as _M
    _M.NARROW_ROW_BYTES = 0
ONLY_H = [int(a) for a in sys.argv[1:]]


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


g = torch.Generator().manual_seed(1)
N0 = 4_000_000
shapes = {
    'uniform U(8,512)': torch.randint(8, 513, (N0 // 260,), generator=g),
    'one giant (2 M) + U(8,64)': torch.cat([torch.randint(8, 65, (N0 // 72,), generator=g), torch.tensor([2_000_000])]),
    '90 % empty': torch.where(torch.rand(150_000, generator=g) < 0.9, torch.tensor(0), torch.randint(8, 513, (150_000,), generator=g)),
    'all length 1': torch.ones(N0, dtype=torch.long),
    'two lengths 1 / 4000': torch.where(torch.rand(40_000, generator=g) < 0.99, torch.tensor(1), torch.tensor(4000)),
}
for H in (ONLY_H or (512, 16)):
    for name, lens in shapes.items():
        N = int(lens.sum())
        data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        cd = ta.C(data, lens.to(dev))
        nb = N * H * 2
        line = f'H={H:3d} {name:28s} N={N:8d} B={lens.numel():7d} T={int(lens.max()):7d} |'
        for op, fn, b in (('pack', lambda: c.pack(), 2 * nb), ('pack(dev lens)', lambda: cd.pack(), 2 * nb), ('P.cat', lambda: p.cat(), 2 * nb),
                          ('P.roll', lambda: p.roll(1), 2 * nb), ('sum(p)', lambda: ta.reduce_sum(p), nb), ('sum(c)', lambda: ta.reduce_sum(c), nb),
                          ('sum(dev lens)', lambda: ta.segment_sum(data, cd.token_sizes), nb), ('max(p)', lambda: ta.reduce_max(p), nb),
                          ('c.ptr', lambda: c.ptr(), 0), ('p.ptr', lambda: p.ptr(), 0)):
            t = med(fn)
            line += f' {op} {t:8.3f} ms' + (f' ({b / t / 1e9:4.2f} TB/s)' if b else '') + ' |'
        print(line, flush=True)
        del data, c, p, cd
        torch.cuda.empty_cache()
