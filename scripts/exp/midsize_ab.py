"""Developer A/B (mid-size BASELINE configs): the row mover's tile size / tile order / cache policy on
the cfg2 shape (B = 4 096, H = 256, bf16), bursts of 8 launches timed with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')


def burst(fn, reps=8, rounds=9):
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


for B, H in ((4096, 256), (4096, 512), (16384, 512)):
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(8, 513, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    cl, pl = describe(c), describe(p)
    out = torch.empty_like(data)
    nb = 2 * N * H * 2
    print(f'--- B={B} H={H} N={N}: {nb / 1e9:.2f} GB per move')
    for name, dst, src, x in (('C->P', pl, cl, data), ('P->C', cl, pl, p.data)):
        for tl in (0, 3, 4, 5, 6):
            for span in (0, 256, 512):
                for nt in (0, 2, 4):
                    flags = (tl << 4) | span | nt
                    us = burst(lambda: O.launch_move(O.MovePlan(dst, src, data.shape, flags=flags), x, out=out))
                    print(f'{name} tile_log2={tl or "auto"} span={ {0: "auto", 256: "on", 512: "off"}[span]:4s} nt={ {0: "auto", 2: "on", 4: "off"}[nt]:4s} '
                          f'{us:8.1f} us  {nb / us / 1e6:5.2f} TB/s', flush=True)
