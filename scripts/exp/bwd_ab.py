"""Developer A/B of the fused reduce backward at the north-star shape, IN ONE PROCESS: every variant (a library build and
/ or knobs) runs on the same tensors, interleaved — buffer placement moves a kernel by +-5 % from process to process
(DESIGN 4.1a), which is as much as the differences looked for.
    python3 scripts/exp/bwd_ab.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L, _meta as M  # noqa: E402


def load_variant(path):
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in L.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


HEAD = L.load()
OLD = os.path.join(ROOT, 'scripts/exp/libs/librua_1826d80.so')
VARIANTS = [('head', HEAD)]          # name, library
for path in os.environ.get('RUA_AB_LIB', OLD).split(','):     # other builds of the library to compare with (make OUT=... BUILD=...)
    if os.path.exists(path):
        VARIANTS.append((os.path.basename(path)[-16:], load_variant(path)))


def use(v):
    L._lib = v[1]


dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
B, H = 65536, 512
lens = torch.randint(8, 513, (B,), generator=g)
n = int(lens.sum())
data = torch.empty((n, H), dtype=torch.bfloat16, device=dev)
for a in range(0, n, 1 << 22):
    data[a:a + (1 << 22)] = torch.randn((min(n, a + (1 << 22)) - a, H), device=dev)
c = ta.with_host_sizes(data, lens)
p = c.pack()
nb = n * H * 2


def once(fn):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


print(f'{"op":16s} ' + ' '.join(f'{v[0]:>16s}' for v in VARIANTS))
for name, passes in (('sum', 1), ('max', 2), ('logsumexp', 2)):
    for tag, z in (('C', c), ('P', p)):
        x = z.data.detach().requires_grad_(True)
        use(VARIANTS[0])
        out = getattr(ta, f'reduce_{name}')(z._replace(data=x))
        cot = torch.randn_like(out)
        fn = lambda: torch.autograd.grad(out, x, cot, retain_graph=True)   # noqa: E731
        ts = {v[0]: [] for v in VARIANTS}
        for v in VARIANTS:
            use(v)
            fn()
        torch.cuda.synchronize()
        for _ in range(5):
            for v in VARIANTS:
                use(v)
                ts[v[0]].append(once(fn))
        med = {k: sorted(t)[len(t) // 2] for k, t in ts.items()}
        print(f'{name + "(" + tag + ")":16s} ' + ' '.join(f'{med[v[0]]:8.3f} {passes * nb / med[v[0]] / 1e9:5.2f}TB' for v in VARIANTS), flush=True)
        del x, out, cot
use(VARIANTS[0])
