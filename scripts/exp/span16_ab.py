"""Developer A/B (late): rows that are a multiple of 16 bytes but not of a 128-byte line (H = 1 000, 328,
1 080 ... in bf16) through the LDS-staged span kernel (move_rows_span_kernel: aligned loads, the 16-KiB destination tile
stored in whole lines) against the row mover (RUA_MOVE_NO_TAIL8 = 1024 turns the span kernel off).  ~8 GB payloads,
HIP events, median of 5; pack, P.cat and C.left (the pad: payload + fill)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
NO_SPAN = 1024        # include/rua.h: RUA_MOVE_NO_TAIL8


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


print(f'{"H":>6} {"row B":>6} | {"pack span":>10} {"TB/s":>5} {"rows":>8} {"TB/s":>5} | {"P.cat span":>10} {"TB/s":>5} {"rows":>8} {"TB/s":>5} | '
      f'{"C.left span":>11} {"TB/s":>5} {"rows":>8} {"TB/s":>5}')
for H in ([int(a) for a in sys.argv[1:]] or (100, 250, 500, 2500, 3000, 4000)):
    rows = int(8e9 / (H * 2))
    B = max(1024, rows // 260)
    g = torch.Generator().manual_seed(H)
    lens = torch.randint(8, 513, (B,), generator=g)
    N, T = int(lens.sum()), int(lens.max())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    l = c.left()
    cl, pl, ll = describe(c), describe(p), describe(l)
    out = torch.empty_like(data)
    outl = torch.empty_like(l.data)
    nb = N * H * 2
    cells = []
    for dst, src, x, o, by in ((pl, cl, data, out, 2 * nb), (cl, pl, p.data, out, 2 * nb), (ll, cl, data, outl, nb + B * T * H * 2)):
        for flags in (0, NO_SPAN):
            t = timeit(lambda: O.launch_move(O.MovePlan(dst, src, o.shape, flags=flags), x, out=o))
            cells.append(f'{t:8.3f} {by / t / 1e9:5.2f}')
    print(f'{H:6d} {H * 2:6d} |   {cells[0]} {cells[1]} |   {cells[2]} {cells[3]} |    {cells[4]} {cells[5]}', flush=True)
    del data, c, p, l, out, outl
    torch.cuda.empty_cache()
