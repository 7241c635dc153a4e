"""Developer probe: where does the HOST time of one pack->reduce step go on the GPU box?"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

print('threads', torch.get_num_threads(), 'cpus', os.cpu_count())
g = torch.Generator().manual_seed(5)
B = int(os.environ.get('RUA_PROBE_B', 65536))
H = int(os.environ.get('RUA_PROBE_H', 512))
lens = torch.randint(8, 513, (B,), generator=g)


def t(fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


for nt in (torch.get_num_threads(), 1):
    torch.set_num_threads(nt)
    print(f'--- {nt} threads')
    print('  sort        %.3f ms' % t(lambda: torch.sort(lens, descending=True)))
    print('  bincount    %.3f ms' % t(lambda: torch.bincount(lens, minlength=513)))
    print('  cumsum      %.3f ms' % t(lambda: torch.cumsum(lens, 0)))
    print('  pin_memory  %.3f ms' % t(lambda: lens.pin_memory()))
    print('  max         %.3f ms' % t(lambda: int(lens.max())))
    print('  sum         %.3f ms' % t(lambda: int(lens.sum())))
    pinned = lens.pin_memory()
    print('  h2d pinned  %.3f ms' % t(lambda: pinned.to('cuda', non_blocking=True)))
    print('  h2d pageable %.3f ms' % t(lambda: lens.to('cuda')))
torch.set_num_threads(os.cpu_count())

dev = torch.device('cuda:0')
data = torch.randn(int(lens.sum()), H, device=dev, dtype=torch.bfloat16)


def step():
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    return ta.reduce_sum(p)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(50):
    step()
host_ms = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
pr.disable()
print('host ms per step (enqueue only)', host_ms)
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
