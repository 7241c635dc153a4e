"""Developer probe: the pack as an INDEX-DRIVEN gather — p.ptr() once (the reference's own formulation: core/cast.py:41-43
is `self[z.ptr()]`), then out = c[batch_ptr, token_ptr] through the mover's LIST path, whose phase 1 is two coalesced loads
and one dependent one instead of the cooperative searches — against c.pack() at the north-star shape.  HIP events, ms."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

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


g = torch.Generator().manual_seed(5)
B, H = 65536, 512
lens = torch.randint(8, 513, (B,), generator=g)
n = int(lens.sum())
data = torch.empty((n, H), dtype=torch.bfloat16, device=dev)
for a in range(0, n, 1 << 22):
    data[a:a + (1 << 22)] = torch.randn((min(n, a + (1 << 22)) - a, H), device=dev)
c = ta.with_host_sizes(data, lens)
p = c.pack()
bp, tp = p.ptr()
out = c[bp, tp]
assert torch.equal(out, p.data)
print(f'c.pack()            {med(lambda: c.pack()):7.3f} ms')
print(f'p.ptr()             {med(lambda: p.ptr()):7.3f} ms')
print(f'c[bp, tp]           {med(lambda: c[bp, tp]):7.3f} ms')
print(f'c.roll(0)           {med(lambda: c.roll(0)):7.3f} ms')
print(f'p.cat()             {med(lambda: p.cat()):7.3f} ms')
cb, ct = c.ptr()
assert torch.equal(p[cb, ct], data)
print(f'p[cb, ct]           {med(lambda: p[cb, ct]):7.3f} ms')
