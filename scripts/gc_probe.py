"""Developer probe: how much of the host time per pack -> reduce step is CPython's cyclic garbage collector?"""
import gc
import os
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


def run(n=2000):
    for _ in range(100):
        full()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        full()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt


stats = {'n': [0, 0, 0], 't': [0.0, 0.0, 0.0], 't0': 0.0}


def cb(phase, info):
    if phase == 'start':
        stats['t0'] = time.perf_counter()
    else:
        stats['n'][info['generation']] += 1
        stats['t'][info['generation']] += time.perf_counter() - stats['t0']


print(f'gc enabled : {run():7.1f} us per step   thresholds {gc.get_threshold()}')
gc.callbacks.append(cb)
dt = run()
gc.callbacks.remove(cb)
print(f'gc enabled : {dt:7.1f} us per step   collections per generation {stats["n"]}, seconds {[round(x, 4) for x in stats["t"]]}'
      f' over 2100 steps')
gc.disable()
print(f'gc disabled: {run():7.1f} us per step')
gc.enable()
gc.freeze()
print(f'gc.freeze(): {run():7.1f} us per step')
print('tracked objects:', len(gc.get_objects()))
