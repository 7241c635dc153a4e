"""Developer A/B of the forward reducers IN ONE PROCESS: HEAD against other builds of the library (RUA_AB_LIB=path[,path]),
interleaved on the same tensors — sum / max / logsumexp at cfg3 (device-only lengths, C; and P), cfg2 over P and the
north-star shape over P; bursts of 8 calls for the short ones, HIP events, us per call.
    RUA_AB_LIB=scripts/exp/librua_base.so python3 scripts/exp/fwd_ab.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L  # noqa: E402


def load_variant(path):
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in L.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


VARIANTS = [('head', L.load())]
for path in os.environ.get('RUA_AB_LIB', '').split(','):
    if path and os.path.exists(path):
        VARIANTS.append((os.path.basename(path)[-16:], load_variant(path)))
dev = torch.device('cuda:0')


def burst(fn, reps, rounds=11):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2] * 1e3


print(f'{"shape":30s} {"op":10s} ' + ' '.join(f'{v[0]:>16s}' for v in VARIANTS))
for tag, B, lo, hi, H, reps in (('cfg3 C device', 16384, 1, 64, 512, 8), ('cfg3 P', 16384, 1, 64, 512, 8), ('cfg2 P', 4096, 8, 512, 256, 8),
                                ('north star P', 65536, 8, 512, 512, 1), ('north star C', 65536, 8, 512, 512, 1)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    n = int(lens.sum())
    data = torch.empty((n, H), dtype=torch.bfloat16, device=dev)
    for a in range(0, n, 1 << 22):
        data[a:a + (1 << 22)] = torch.randn((min(n, a + (1 << 22)) - a, H), device=dev)
    host = ta.with_host_sizes(data, lens)
    z = ta.C(data, lens.to(dev)) if tag.endswith('device') else (host.pack() if tag.endswith('P') else host)
    for name in ('sum', 'max', 'logsumexp'):
        fn = getattr(ta, f'reduce_{name}')
        cells = []
        for v in VARIANTS:
            L._lib = v[1]
            cells.append(f'{burst(lambda: fn(z), reps):16.1f}')
        L._lib = VARIANTS[0][1]
        print(f'{tag:30s} {name:10s} ' + ' '.join(cells), flush=True)
    del data, host, z
    torch.cuda.empty_cache()
