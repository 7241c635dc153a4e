"""Developer probe: the selects, gathers and enumerations on degenerate length distributions against a uniform one of the
same size (bf16, H = 64): head / last / rev / trunc / C.roll, X[batch_ptr, token_ptr], idx(), ptr(), masks on small T."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


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
    'all length 3': torch.full((N0 // 3,), 3),
}
H = 64
for name, lens in shapes.items():
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    cd = ta.C(data, lens.to(dev))
    bp, tp = c.ptr()
    line = f'{name:28s} N={N:8d} B={lens.numel():7d} |'
    ops = [('c.rev', lambda: c.rev()), ('p.rev', lambda: p.rev()), ('c.roll(1)', lambda: c.roll(1)), ('c.head(1)', lambda: c.head(1)),
           ('p.head(1)', lambda: p.head(1)), ('c.last', lambda: c.last()), ('p.last', lambda: p.last()),
           ('c.trunc((1,1))', lambda: c.trunc((1, 1))), ('p.trunc((1,1))', lambda: p.trunc((1, 1))),
           ('c.idx', lambda: c.idx()), ('p.idx', lambda: p.idx()), ('cd.ptr', lambda: cd.ptr()),
           ('p[bp,tp]', lambda: p[bp, tp]), ('c.cat_view', lambda: c.cat())]
    for op, fn in ops:
        try:
            t = med(fn)
            line += f' {op} {t:7.3f} |'
        except Exception as e:      # (e.g. a trunc that empties every sequence)
            line += f' {op} {type(e).__name__} |'
    print(line, flush=True)
    del data, c, p, cd
    torch.cuda.empty_cache()
