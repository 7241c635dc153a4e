"""Developer A/B: launch geometry of the row mover (rows per workgroup tile x tile order) at the north-star shape,
interleaved in one process.  Flags: include/rua.h RUA_MOVE_TILE_LOG2 / RUA_MOVE_XCD_SPAN_ON."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
B, H = int(os.environ.get('B', 65536)), int(os.environ.get('H', 512))
LO, HI = int(os.environ.get('LO', 8)), int(os.environ.get('HI', 512))
g = torch.Generator().manual_seed(5)
lens = torch.randint(LO, HI + 1, (B,), generator=g)
N = int(lens.sum())
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
c = ta.with_host_sizes(data, lens)
p = c.pack()
rb = H * 2
cl, pl = describe(c), describe(p)
out = torch.empty_like(data)
S = L.stream_ptr(dev)
SPAN_ON, SPAN_OFF = 256, 512


def mover(kind, flags):
    d, s_, src = {'C->P': (pl, cl, data), 'P->C': (cl, pl, p.data), 'C->C': (cl, cl, data)}[kind]
    return lambda: L.check(lib.rua_move_rows(d.ref(), s_.ref(), 0, 0, out.data_ptr(), src.data_ptr(), rb, None, -1, flags, S), 'm')


variants = {}
kinds = os.environ.get('KINDS', 'C->P,P->C,C->C').split(',')
for kind in kinds:
    variants[f'{kind} default policy'] = (mover(kind, 0), kind)
    for k in (2, 3, 4, 5, 6, 8):
        for span in (False, True):
            variants[f'{kind} tile {1 << k:3d} rows {"span/XCD" if span else "linear  "}'] = (mover(kind, (k << 4) | (SPAN_ON if span else SPAN_OFF)), kind)
expect = {'C->P': p.data, 'P->C': data, 'C->C': data}
for name, (fn, kind) in variants.items():
    out.zero_()
    fn()
    torch.cuda.synchronize()
    assert torch.equal(out, expect[kind]), name
times = {k: [] for k in variants}
for rnd in range(7):
    for name, (fn, _) in variants.items():
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
nbytes = 2 * N * rb
print(f'B={B} H={H} len~U({LO},{HI}) N={N} rows of {rb} B; {nbytes / 1e9:.2f} GB per move')
for name, ts in times.items():
    ts = sorted(ts)
    print(f'{name:36s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  {nbytes / ts[len(ts) // 2] / 1e9:.2f} TB/s')
