"""Developer probe: where do host stalls in the north-star step loop come from (GC? allocator? ring?)."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')
B, H = 65536, 512
g = torch.Generator().manual_seed(5)
lens = torch.randint(8, 513, (B,), generator=g)
n = int(lens.sum())
data = torch.empty((n, H), dtype=torch.bfloat16, device=dev).normal_()

events = []
t_gc = [0.0]


def cb(phase, info):
    if phase == 'start':
        t_gc[0] = time.perf_counter()
    else:
        events.append(('gc', info['generation'], (time.perf_counter() - t_gc[0]) * 1e3, info.get('collected')))


gc.callbacks.append(cb)
mode = sys.argv[1] if len(sys.argv) > 1 else 'copy'
p = out = None
for i in range(30):
    t0 = time.perf_counter()
    if mode == 'copy':
        c = ta.with_host_sizes(data, lens)
    else:
        dsz = M.to_device_async(lens, dev)
        M.attach_host(dsz, lens)
        c = ta.C(data, dsz)
    t1 = time.perf_counter()
    p = c.pack()
    t2 = time.perf_counter()
    out = ta.reduce_sum(p)
    t3 = time.perf_counter()
    print(f'step {i:2d}: new {1e3 * (t1 - t0):7.2f}  pack {1e3 * (t2 - t1):7.2f}  reduce {1e3 * (t3 - t2):7.2f}   {events}')
    events.clear()
torch.cuda.synchronize()
