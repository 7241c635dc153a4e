"""Developer probe: what does the mover's row resolution (phase 1) still cost?  The same 34.9 GB move three ways:
C -> C with real ragged lengths (64-ary search of `off`), C -> C with CONSTANT lengths (closed form: no index load at
all), C -> P (search of `boff`, then sorted / off gathers)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L, _meta as M  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
B, H = 65536, 512
g = torch.Generator().manual_seed(5)
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
out = torch.empty_like(data)
c = ta.with_host_sizes(data, lens)
p = c.pack()
cl, pl = describe(c), describe(p)
Lc = 256
Bc = N // Lc
const = M.lay_cat(None, Bc, Bc * Lc, len_add=Lc)
S = L.stream_ptr(dev)
rb = H * 2


def run(d, s_, src):
    return lambda: L.check(lib.rua_move_rows(d.ref(), s_.ref(), 0, 0, out.data_ptr(), src.data_ptr(), rb, None, -1, 0, S), 'm')


variants = {'C->C ragged (64-ary search of off)': (run(cl, cl, data), N), 'C->C constant lengths (no index loads)': (run(const, const, data), Bc * Lc),
            'C->P (boff search + sorted/off gathers)': (run(pl, cl, data), N), 'P->C': (run(cl, pl, p.data), N)}
times = {k: [] for k in variants}
for fn, _ in variants.values():
    fn()
torch.cuda.synchronize()
for rnd in range(7):
    for name, (fn, _) in variants.items():
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
for name, ts in times.items():
    ts = sorted(ts)
    nb = 2 * variants[name][1] * rb
    print(f'{name:44s} median {ts[len(ts) // 2]:.3f} ms  {nb / ts[len(ts) // 2] / 1e9:.2f} TB/s')
