"""Developer A/B: C -> P at the north-star shape, destination-ordered (a tile = 16 consecutive P rows, gathered from 16
sequences) against source-ordered (a tile = 16 consecutive C rows, scattered to 16 time steps: RUA_MOVE_SCATTER)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
B, H = 65536, int(os.environ.get('H', 512))
g = torch.Generator().manual_seed(5)
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
c = ta.with_host_sizes(data, lens)
p = c.pack()
rb = H * 2
cl, pl = describe(c), describe(p)
out = torch.empty_like(data)
S = L.stream_ptr(dev)
SCATTER = 1
variants = {
    'C->P gather  (tiles of P rows)': (lambda: L.check(lib.rua_move_rows(pl.ref(), cl.ref(), 0, 0, out.data_ptr(), data.data_ptr(), rb, None, -1, 0, S), 'm'), p.data),
    'C->P scatter (tiles of C rows)': (lambda: L.check(lib.rua_move_rows(cl.ref(), pl.ref(), 0, 0, out.data_ptr(), data.data_ptr(), rb, None, -1, SCATTER, S), 'm'), p.data),
    'P->C gather  (tiles of C rows)': (lambda: L.check(lib.rua_move_rows(cl.ref(), pl.ref(), 0, 0, out.data_ptr(), p.data.data_ptr(), rb, None, -1, 0, S), 'm'), data),
    'P->C scatter (tiles of P rows)': (lambda: L.check(lib.rua_move_rows(pl.ref(), cl.ref(), 0, 0, out.data_ptr(), p.data.data_ptr(), rb, None, -1, SCATTER, S), 'm'), data),
}
for name, (fn, want) in variants.items():
    out.zero_()
    fn()
    torch.cuda.synchronize()
    assert torch.equal(out, want), name
times = {k: [] for k in variants}
for rnd in range(9):
    for name, (fn, _) in variants.items():
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
nbytes = 2 * N * rb
for name, ts in times.items():
    ts = sorted(ts)
    print(f'{name:34s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  {nbytes / ts[len(ts) // 2] / 1e9:.2f} TB/s')
