"""Developer A/B (one process): tile shapes for moves OUT of a PackedSequence at 8- and 4-byte rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')
TALL = dict(M._FROM_PACK_SHAPES)


def once(fn):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for dtype in (torch.int64, torch.float32):
    g = torch.Generator().manual_seed(8)
    B = 1923076
    lens = torch.randint(8, 513, (B,), generator=g)
    n = int(lens.sum())
    data = torch.randint(0, 100, (n,), device=dev, dtype=torch.int32).to(dtype)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = n * data.element_size()
    T = int(lens.max())
    def repack():
        M.forget(c.token_sizes)
        return ta.with_host_sizes(data, lens).pack()

    for name, fn, alg in (('P.cat', lambda: p.cat(), 2 * nb), ('P.left', lambda: p.left(), nb + B * T * data.element_size()),
                          ('pack', lambda: c.pack(), 2 * nb)):
        res = {}
        for tag, shapes in (('tall', TALL), ('square', {})):
            M._FROM_PACK_SHAPES = shapes
            M._TALL_BOTH_WAYS = True
            fn()
        torch.cuda.synchronize()
        for tag, shapes in (('tall', TALL), ('square', {})) * 5:
            M._FROM_PACK_SHAPES = shapes
            M._TALL_BOTH_WAYS = True
            res.setdefault(tag, []).append(once(fn))
        print(f'{str(dtype):14s} {name:7s} ' + '  '.join(f'{k}: {sorted(v)[2]:.3f} ms {alg / sorted(v)[2] / 1e9:.2f} TB/s' for k, v in res.items()), flush=True)
    del data, c, p
M._FROM_PACK_SHAPES = TALL
M._TALL_BOTH_WAYS = False
