"""Developer probe: row widths that are not powers of two (H = 320, 384, 768, 1000 bf16) — pack, P.cat, pad, reduce."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


print(f'{"H":>6} {"row B":>6} | {"pack":>8} {"TB/s":>6} | {"P.cat":>8} {"TB/s":>6} | {"C.left":>8} {"TB/s":>6} | {"reduce(P)":>9} {"TB/s":>6} | {"seg_max(C)":>10} {"TB/s":>6}')
for H in (320, 384, 500, 768, 1000, 1536):
    rows = int(8e9 / (H * 2))
    B = max(1024, rows // 260)
    g = torch.Generator().manual_seed(H)
    lens = torch.randint(8, 513, (B,), generator=g)
    N = int(lens.sum())
    T = int(lens.max())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = N * H * 2
    t_pack = timeit(lambda: c.pack())
    t_cat = timeit(lambda: p.cat())
    t_left = timeit(lambda: c.left())
    t_red = timeit(lambda: ta.reduce_sum(p))
    t_max = timeit(lambda: ta.segment_max(c.data, c.token_sizes))
    print(f'{H:6d} {H * 2:6d} | {t_pack:8.3f} {2 * nb / t_pack / 1e9:6.2f} | {t_cat:8.3f} {2 * nb / t_cat / 1e9:6.2f} | '
          f'{t_left:8.3f} {(nb + B * T * H * 2) / t_left / 1e9:6.2f} | {t_red:9.3f} {nb / t_red / 1e9:6.2f} | {t_max:10.3f} {nb / t_max / 1e9:6.2f}')
    del data, c, p
