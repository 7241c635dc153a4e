"""Developer bench: pack / reduce / pad kernels across row widths at a fixed payload size (~8 GB)."""
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


print(f'{"H":>6} {"row B":>6} {"B":>8} {"N":>10} | {"pack ms":>8} {"TB/s":>6} | {"P.cat ms":>8} {"TB/s":>6} | {"reduce(P)":>9} {"TB/s":>6} | {"seg_sum":>8} {"TB/s":>6} | {"roll(P)":>8} {"TB/s":>6}')
for H in ([int(a) for a in sys.argv[1:]] or (16, 32, 64, 128, 256, 512, 1024, 2048)):
    rows = int(8e9 / (H * 2))
    B = max(1024, rows // 260)
    g = torch.Generator().manual_seed(H)
    lens = torch.randint(8, 513, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = N * H * 2
    from torchrua_amd import _ops as O
    from torchrua_amd.layout import describe
    from torchrua_amd import _meta as M
    cl, pl = describe(c), describe(p)
    out = torch.empty_like(data)
    t_pack = timeit(lambda: O.launch_move(O.MovePlan(pl, cl, data.shape), data, out=out))
    t_cat = timeit(lambda: O.launch_move(O.MovePlan(cl, pl, data.shape), p.data, out=out))
    t_roll = timeit(lambda: O.launch_move(O.MovePlan(pl, pl, data.shape, tmap=1, arg=1), p.data, out=out))
    t_red = timeit(lambda: ta.reduce_sum(p))
    t_seg = timeit(lambda: ta.segment_sum(c.data, c.token_sizes))
    print(f'{H:6d} {H * 2:6d} {B:8d} {N:10d} | {t_pack:8.3f} {2 * nb / t_pack / 1e9:6.2f} | {t_cat:8.3f} {2 * nb / t_cat / 1e9:6.2f} | '
          f'{t_red:9.3f} {nb / t_red / 1e9:6.2f} | {t_seg:8.3f} {nb / t_seg / 1e9:6.2f} | {t_roll:8.3f} {2 * nb / t_roll / 1e9:6.2f}')
    del data, c, p, out
