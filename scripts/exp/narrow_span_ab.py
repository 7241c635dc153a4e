"""Developer A/B: one contiguous span of tiles per XCD vs plain blockIdx order for the (rank x time) tile kernels at
narrow rows (pack, P.cat, roll inside a PackedSequence), 8 GB payloads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

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


payload = float(sys.argv[1]) if len(sys.argv) > 1 else 8e9
for H in (8, 16, 32):
    rows = int(payload / (H * 2))
    B = max(1024, rows // 260)
    lens = torch.randint(8, 513, (B,), generator=torch.Generator().manual_seed(H))
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    cl, pl = describe(c), describe(p)
    out = torch.empty_like(data)
    nb = 2 * N * H * 2
    line = f'row {H * 2:3d} B  N={N:10d} tiles={pl.c.n_tiles:8d} |'
    for name, dst, src, x, kw in (('pack', pl, cl, data, {}), ('P.cat', cl, pl, p.data, {}),
                                  ('roll', pl, pl, p.data, dict(tmap=1, arg=1))):
        t = {}
        for rep in range(2):
            for span in (256, 512, 0):
                t.setdefault(span, []).append(med(lambda: O.launch_move(O.MovePlan(dst, src, data.shape, flags=span, **kw), x, out=out)))
        f = lambda s: min(t[s])
        line += f' {name}: on {f(256):6.3f} off {f(512):6.3f} auto {f(0):6.3f} ms ({nb / f(256) / 1e9:4.2f} / {nb / f(512) / 1e9:4.2f} TB/s) |'
    print(line, flush=True)
    del data, c, p, out
    torch.cuda.empty_cache()
