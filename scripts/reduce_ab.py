"""Developer A/B: the reducer across build variants (librua_hip_<tag>.so next to the library, e.g. built with
EXTRA=-DRUA_UNROLL_T=4), interleaved in one process: reduce over P and C at the north-star shape, cfg2, cfg3."""
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
S = L.stream_ptr(dev)


def case(tag, B, lo, hi, H, op):
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    n = int(lens.sum())
    data = torch.randn(n, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    out = torch.empty(B, H, device=dev, dtype=torch.bfloat16)
    res = {}
    for kind, z in (('P', p), ('C', c)):
        lay = describe(z)
        fns = {t: (lambda lib=lib: L.check(lib.rua_segment_reduce(lay.ref(), None, z.data.data_ptr(), out.data_ptr(), H, L.BF16, op, 0, 0, None, 0, None, None, S), 'r'))
               for t, lib in libs.items()}
        times = {t: [] for t in fns}
        for fn in fns.values():
            fn()
        torch.cuda.synchronize()
        for rnd in range(7):
            for t, fn in fns.items():
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / 4)
        nb = n * H * 2 + B * H * 2
        res[kind] = '  '.join(f'{t}: {sorted(ts)[3] * 1e3:8.1f} us {nb / sorted(ts)[3] / 1e9:5.2f} TB/s' for t, ts in times.items())
    for kind, line in res.items():
        print(f'{tag:12s} op {op} over {kind}:  {line}')


case('north star', 65536, 8, 512, 512, L.SUM)
case('north star', 65536, 8, 512, 512, L.LOGSUMEXP)
case('cfg2', 4096, 8, 512, 256, L.SUM)
case('cfg3', 16384, 1, 64, 512, L.SUM)
case('cfg3', 16384, 1, 64, 512, L.MAX)
case('cfg4-ish', 16384, 16, 1024, 1024, L.SUM)
