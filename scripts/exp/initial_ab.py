"""Developer probe: what tracking the reference's global `initial` inside the reduce costs — max and
logsumexp against sum (which has no `initial`, no scratch, no trailing launch) at cfg3, cfg2 and three shapes of many
short sequences; bursts of 8 calls, HIP events, us per call; bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def burst(fn, reps=8, rounds=15):
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


print(f'{"shape":32s} {"layout":10s} {"sum us":>8s} {"max us":>8s} {"lse us":>8s}')
for tag, B, lo, hi, H in (('cfg3 16384 x U(1,64) H=512', 16384, 1, 64, 512), ('cfg2 4096 x U(8,512) H=256', 4096, 8, 512, 256),
                          ('65536 x U(1,16) H=64', 65536, 1, 16, 64), ('200000 x U(1,32) H=8', 200000, 1, 32, 8),
                          ('1000000 x U(1,8) H=32', 1000000, 1, 8, 32)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    host = ta.with_host_sizes(data, lens)
    for lname, z in (('C device', ta.C(data, lens.to(dev))), ('C host', host), ('P', host.pack())):
        ts = [burst(lambda: getattr(ta, f'reduce_{name}')(z)) for name in ('sum', 'max', 'logsumexp')]
        print(f'{tag:32s} {lname:10s} {ts[0]:8.1f} {ts[1]:8.1f} {ts[2]:8.1f}', flush=True)
