"""Developer probe: the north-star pipeline step by step with and without placement-aware outputs (RUA_PLACEMENT),
in ONE process is impossible (the allocator state differs) — run this script once per setting and compare the per-step
pack kernel times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops, _placement  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, H = 65536, 512
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
pairs = {'to_pack': [], 'reduce': []}
state = {}


def hook(name, start):
    if name not in pairs:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    if start:
        state[name] = e
    else:
        pairs[name].append((state[name], e))


p = out = None
t_first = []
for i in range(12):
    t0 = time.perf_counter()
    p, out = (lambda c: (lambda q: (q, ta.reduce_sum(q)))(c.pack()))(ta.with_host_sizes(data, lens))
    torch.cuda.synchronize()
    t_first.append((time.perf_counter() - t0) * 1e3)
print('first twelve steps, wall ms each (probes happen here):', ' '.join(f'{x:.1f}' for x in t_first))
_ops.set_kernel_hook(hook)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 24
host = []
for i in range(K):
    h0 = time.perf_counter()
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    out = ta.reduce_sum(p)
    host.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
print('  host ms per step:', ' '.join(f'{x:.2f}' for x in host))
dt = (time.perf_counter() - t0) / K * 1e3
print(f'RUA_PLACEMENT={os.environ.get("RUA_PLACEMENT", "1")}: {dt:.3f} ms/step  {N * H / dt / 1e3:.0f} M elements/s')
for name in pairs:
    print(f'  {name} kernel ms:', ' '.join(f'{a.elapsed_time(b):.2f}' for a, b in pairs[name]))
print('  placement stats:', _placement.stats, ' reserved GB:', round(torch.cuda.memory_reserved() / 1e9, 1))
