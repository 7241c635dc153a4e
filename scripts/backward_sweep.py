"""Developer bench: the fused reduce backward over C and P across row widths (~4 GB payload).
Algorithmic bytes: sum/mean write N*H*e; max/logsumexp read the payload and write the gradient (the forward counted max's ties)."""
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


print(f'{"H":>6} {"row B":>6} | ' + ' | '.join(f'{k + "(" + z + ")":>14} {"TB/s":>5}' for k in ('sum', 'max', 'lse') for z in 'CP'))
for H in (16, 32, 64, 128, 512, 2048):
    rows = int(4e9 / (H * 2))
    B = max(1024, rows // 260)
    g = torch.Generator().manual_seed(H)
    lens = torch.randint(8, 513, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    nb = N * H * 2
    cells = []
    for name, passes in (('sum', 1), ('max', 2), ('logsumexp', 2)):
        for z in (c, p):
            x = z.data.detach().requires_grad_(True)
            out = getattr(ta, f'reduce_{name}')(z._replace(data=x))
            cot = torch.ones_like(out)
            t = timeit(lambda: torch.autograd.grad(out, x, cot, retain_graph=True))
            cells.append(f'{t:14.3f} {passes * nb / t / 1e9:5.2f}')
            del x, out
    print(f'{H:6d} {H * 2:6d} | ' + ' | '.join(cells), flush=True)
    del data, c, p
