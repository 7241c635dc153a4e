"""Developer probe: cProfile of the host side of with_host_sizes -> pack -> reduce_sum at B = 4096 (tiny payload, so the
GPU never back-pressures the launches)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
lens = torch.randint(1, 5, (4096,), generator=g)
data = torch.randn(int(lens.sum()), 256, device=dev, dtype=torch.bfloat16)


def full():
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    return ta.reduce_sum(p)


for _ in range(100):
    full()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    full()
dt = (time.perf_counter() - t0) / 500 * 1e6
torch.cuda.synchronize()
print(f'host per step: {dt:.1f} us')
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    full()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(32)
pstats.Stats(pr).print_callers('__new__')
