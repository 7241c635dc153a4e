"""Developer probe: is the narrow-row P.cat / pack gap an ALIGNMENT cost?  Same payload, (a) ragged U(8,512) lengths (runs of a
time step start at arbitrary 32-byte offsets), (b) every sequence 256 rows and B a multiple of 16 (every run of both layouts
starts on a 128-byte line), (c) as (b) with B = 16 k + 1 (packed runs misaligned by one row per step, batch-major runs aligned)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')


def med(fn, rounds=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for H in (16, 8, 32):
    rows = int(8e9 / (H * 2))
    for name, lens in (('ragged U(8,512)', torch.randint(8, 513, (rows // 260,), generator=torch.Generator().manual_seed(H))),
                       ('all 256, B = 16 k', torch.full(((rows // 256) // 16 * 16,), 256)),
                       ('all 256, B = 16 k + 1', torch.full(((rows // 256) // 16 * 16 + 1,), 256)),
                       ('all 255, B = 16 k', torch.full(((rows // 256) // 16 * 16,), 255))):
        N = int(lens.sum())
        data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        cl, pl = describe(c), describe(p)
        out = torch.empty_like(data)
        nb = 2 * N * H * 2
        t_pack = med(lambda: O.launch_move(O.MovePlan(pl, cl, data.shape), data, out=out))
        t_cat = med(lambda: O.launch_move(O.MovePlan(cl, pl, data.shape), p.data, out=out))
        t_roll = med(lambda: O.launch_move(O.MovePlan(pl, pl, data.shape, tmap=1, arg=1), p.data, out=out))
        print(f'{H * 2:3d}-byte rows {name:24s}: pack {t_pack:6.3f} ms ({nb / t_pack / 1e9:4.2f} TB/s)  P.cat {t_cat:6.3f} ({nb / t_cat / 1e9:4.2f})  P.roll {t_roll:6.3f} ({nb / t_roll / 1e9:4.2f})', flush=True)
        del data, c, p, out
        torch.cuda.empty_cache()
