"""Developer probe: reduce + pack backward at the mid-size BASELINE shapes (wall per call incl. the Python enqueue,
HIP events, GPU kept busy by back-to-back calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, lo, hi, H) in ((512, 8, 512, 512), (4096, 8, 512, 256), (8192, 8, 512, 512), (16384, 1, 64, 512)):
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    nb = N * H * 2
    row = [f'B={B:6d} H={H:4d} {nb / 1e6:6.0f} MB']
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    for name, passes in (('sum', 1), ('max', 2), ('logsumexp', 2)):
        for z, tag in ((c, 'C'), (p, 'P')):
            x = z.data.detach().requires_grad_(True)
            out = getattr(ta, f'reduce_{name}')(z._replace(data=x))
            cot = torch.ones_like(out)
            us = timeit(lambda: torch.autograd.grad(out, x, cot, retain_graph=True))
            row.append(f'{name}({tag}) {us:7.1f} us {passes * nb / us / 1e6:5.2f}')
    x = c.data.detach().requires_grad_(True)
    pk = ta.C(x, c.token_sizes).pack()
    cot = torch.ones_like(pk.data)
    us = timeit(lambda: torch.autograd.grad(pk.data, x, cot, retain_graph=True))
    row.append(f'pack^T {us:7.1f} us {2 * nb / us / 1e6:5.2f}')
    print(' | '.join(row), flush=True)
