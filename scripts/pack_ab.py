"""Developer A/B (not a test): C->P and P->C at the north-star shape, generic row-major tiles vs the
(rank x time) tiled kernel, across build variants, interleaved in ONE process (cdna guide §5.4 rule 24)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L, _meta as M  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
here = os.path.dirname(L.LIB_PATH)
libs = {'base': L.load()}
for tag in ('nt', 'u8'):
    path = os.path.join(here, f'librua_hip_{tag}.so')
    if os.path.exists(path):
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
assert torch.equal(p.cat().data, data)
rb = H * 2
cl, pl = describe(c), describe(p)
t = M.pack_tiling(p)
out = torch.empty_like(data)
S = L.stream_ptr(dev)


def generic(lib, to_pack):
    d, s_, src = (pl, cl, data) if to_pack else (cl, pl, p.data)
    return lambda: L.check(lib.rua_move_rows(d.ref(), s_.ref(), 0, 0, out.data_ptr(), src.data_ptr(), rb, None, -1, 0, S), 'm')


def tiled(lib, to_pack):
    src = data if to_pack else p.data
    return lambda: L.check(lib.rua_pack_rows(pl.ref(), cl.ref(), int(to_pack), out.data_ptr(), src.data_ptr(), rb,
                                             t.bsz.data_ptr(), t.tile_start.data_ptr(), t.n_chunks, t.n_tiles, S), 'p')


variants = {}
for tag, lib in libs.items():
    for dname, to_pack in (('C->P', True), ('P->C', False)):
        variants[f'{dname} generic {tag}'] = (generic(lib, to_pack), to_pack)
        variants[f'{dname} tiled   {tag}'] = (tiled(lib, to_pack), to_pack)

variants['plain copy (torch copy_)'] = (lambda: out.copy_(data), None)
for name, (fn, to_pack) in variants.items():   # correctness of every variant first
    if to_pack is None:
        continue
    out.zero_()
    fn()
    torch.cuda.synchronize()
    assert torch.equal(out, p.data if to_pack else data), name

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
for name, ts in times.items():
    ts = sorted(ts)
    print(f'{name:24s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  {nbytes / ts[len(ts) // 2] / 1e9:.2f} TB/s')
