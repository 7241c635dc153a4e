"""Developer probe: the fused reduce backward over batches of SHORT sequences at narrow rows (where the forward went
side by side in a wave): sum / max / logsumexp over C (host-known lengths) and P; HIP events, us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def med(fn, rounds=9):
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
    return sorted(ts)[len(ts) // 2] * 1e3


print(f'{"shape":28s} {"layout":8s} {"sum us":>8s} {"max us":>8s} {"lse us":>8s}')
for tag, B, lo, hi, H in (('200000 x U(1,32) H=8', 200000, 1, 32, 8), ('65536 x U(1,16) H=64', 65536, 1, 16, 64),
                          ('1000000 x U(1,8) H=32', 1000000, 1, 8, 32), ('40000 x U(1,100) H=8', 40000, 1, 100, 8)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn(int(lens.sum()), H, device=dev, dtype=torch.float32)
    c = ta.with_host_sizes(data, lens)
    for lname, z in (('C host', c), ('P', c.pack())):
        ts = []
        for name in ('sum', 'max', 'logsumexp'):
            x = z.data.detach().requires_grad_(True)
            out = getattr(ta, f'reduce_{name}')(z._replace(data=x))
            cot = torch.ones_like(out)
            ts.append(med(lambda: torch.autograd.grad(out, x, cot, retain_graph=True)))
        print(f'{tag:28s} {lname:8s} {ts[0]:8.1f} {ts[1]:8.1f} {ts[2]:8.1f}', flush=True)
