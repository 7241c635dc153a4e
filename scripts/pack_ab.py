"""Developer A/B (not a test): the C->P / P->C mover at the north-star shape across build variants
(librua_hip_<tag>.so next to the library), interleaved in ONE process (cdna guide §5.4 rule 24)."""
import ctypes
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
here = os.path.dirname(L.LIB_PATH)
libs = {'base': L.load()}
for path in sorted(glob.glob(os.path.join(here, 'librua_hip_*.so'))):
    tag = os.path.basename(path)[len('librua_hip_'):-3]
    lib = ctypes.CDLL(path)
    for name, (res, args) in L.SYMBOLS.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    libs[tag] = lib

B, H = int(os.environ.get('B', 65536)), int(os.environ.get('H', 512))
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


def mover(lib, to_pack):
    d, s_, src = (pl, cl, data) if to_pack else (cl, pl, p.data)
    return lambda: L.check(lib.rua_move_rows(d.ref(), s_.ref(), 0, 0, out.data_ptr(), src.data_ptr(), rb, None, -1, 0, S), 'm')


variants = {}
for tag, lib in libs.items():
    variants[f'C->P {tag}'] = (mover(lib, True), True)
    variants[f'P->C {tag}'] = (mover(lib, False), False)
for name, (fn, to_pack) in variants.items():
    out.zero_()
    fn()
    torch.cuda.synchronize()
    assert torch.equal(out, p.data if to_pack else data), name
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
for name, ts in sorted(times.items()):
    ts = sorted(ts)
    print(f'{name:18s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  {nbytes / ts[len(ts) // 2] / 1e9:.2f} TB/s')
