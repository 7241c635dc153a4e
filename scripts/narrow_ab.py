"""Developer A/B: narrow rows (32-256 B) between C and P — the (rank x time) tile kernels (default for rows <= 128 B)
against the generic row mover with its round-2 geometry (any geometry flag bypasses the tile kernels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
S = L.stream_ptr(dev)
SPAN_ON, SPAN_OFF = 256, 512
for H in (16, 32, 64):
    rows = int(4e9 / (H * 2))
    B = max(1024, rows // 260)
    g = torch.Generator().manual_seed(H)
    lens = torch.randint(8, 513, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    cl, pl = describe(c), describe(p)
    out = torch.empty_like(data)
    rb = H * 2

    def mover(kind, flags):
        d, s_, src = (pl, cl, data) if kind == 'C->P' else (cl, pl, p.data)
        return lambda: L.check(lib.rua_move_rows(d.ref(), s_.ref(), 0, 0, out.data_ptr(), src.data_ptr(), rb, None, -1, flags, S), 'm')

    variants = {}
    for kind in ('C->P', 'P->C'):
        variants[f'{kind} default'] = (mover(kind, 0), kind)
        variants[f'{kind} flags: span (tiles <= 64 B, else generic)'] = (mover(kind, SPAN_ON), kind)
        variants[f'{kind} flags: linear'] = (mover(kind, SPAN_OFF), kind)
        for k in (6, 7):
            variants[f'{kind} generic, tile {1 << k} rows span'] = (mover(kind, SPAN_ON | (k << 4)), kind)
    expect = {'C->P': p.data, 'P->C': data}
    times = {k: [] for k in variants}
    for name, (fn, kind) in variants.items():
        out.zero_()
        fn()
        torch.cuda.synchronize()
        assert torch.equal(out, expect[kind]), name
    for rnd in range(5):
        for name, (fn, _) in variants.items():
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 3)
    print(f'--- rows of {rb} B, B={B}, N={N}')
    for name, ts in times.items():
        ts = sorted(ts)
        print(f'  {name:40s} {ts[len(ts) // 2]:8.3f} ms  {2 * N * rb / ts[len(ts) // 2] / 1e9:5.2f} TB/s')
    del data, c, p, out
