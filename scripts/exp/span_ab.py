"""Developer A/B: one contiguous span of tiles per XCD vs plain blockIdx order, by destination size (C->P, P->C, C->L)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as K  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')


def med(fn, reps, rounds=7):
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


for B, H in ([(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else ((8192, 512), (16384, 512), (32768, 512), (65536, 512), (32768, 1024))):
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(8, 513, (B,), generator=g)
    N = int(lens.sum())
    data = torch.empty(N, H, device=dev, dtype=torch.bfloat16).normal_()
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    cl, pl = describe(c), describe(p)
    out = torch.empty_like(data)
    T = int(lens.max())
    ll = M.lay_padded(K.LEFT, c.token_sizes, B, T, T)
    pad = torch.empty(B, T, H, device=dev, dtype=torch.bfloat16)
    nb = 2 * N * H * 2
    reps = 8 if nb < 4e9 else 2
    line = f'B={B:6d} H={H:5d} dst={nb / 2e9:6.2f} GB tiles={N * H * 2 // 16384:8d} |'
    for name, dst, src, x, o, bytes_ in (('C->P', pl, cl, data, out, nb), ('P->C', cl, pl, p.data, out, nb),
                                          ('C->L', ll, cl, data, pad, N * H * 2 + B * T * H * 2)):
        t = {}
        for span in (256, 512):
            t[span] = med(lambda: O.launch_move(O.MovePlan(dst, src, o.shape, flags=span), x, out=o), reps)
        line += f' {name}: on {t[256]:8.1f} off {t[512]:8.1f} us ({(t[512] / t[256] - 1) * 100:+5.1f} %) |'
    print(line, flush=True)
    del data, c, p, out, pad
    torch.cuda.empty_cache()
