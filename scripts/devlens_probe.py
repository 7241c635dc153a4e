"""Developer probe: the reference-signature step `C(data, token_sizes_on_device).pack() -> reduce_sum` at the north-star
shape — where the GPU idles between one step's reduce and the next step's mover, piece by piece on the host."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd import core  # noqa: E402

dev = torch.device('cuda:0')
B, H = 65536, 512
g = torch.Generator().manual_seed(5)
lens = torch.randint(8, 513, (B,), generator=g)
n = int(lens.sum())
data = torch.empty((n, H), dtype=torch.bfloat16, device=dev).normal_()

acc = {}


def timed(mod, name):
    fn = getattr(mod, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    setattr(mod, name, wrap)


for name in ('_read_back', 'sorted_indices_to_device', 'batch_sizes_from_host_lens', 'host_sort_desc'):
    timed(M, name)
timed(O, 'launch_move')
timed(O, 'launch_reduce')
timed(core, '_pack_meta')

gap_pairs = []
marks = {}


def hook(name, begin):
    if name == 'to_pack' and begin:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks['pack_begin'] = ev
        if 'reduce_end' in marks:
            gap_pairs.append((marks['reduce_end'], ev))
    if name == 'reduce' and not begin:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks['reduce_end'] = ev


O.set_kernel_hook(hook)
mode = sys.argv[1] if len(sys.argv) > 1 else 'dev'
steps = 24
p = out = None
rows = []
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(steps):
    acc.clear()
    t0 = time.perf_counter()
    if mode == 'dev':
        c = ta.C(data, lens.to(dev))
    else:
        c = ta.with_host_sizes(data, lens)
    t1 = time.perf_counter()
    p = c.pack()
    t2 = time.perf_counter()
    out = ta.reduce_sum(p)
    t3 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), dict(acc)))
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / steps * 1e3
print(f'mode={mode}: {wall:.3f} ms/step wall')
for i, (a, b, c_, d) in enumerate(rows[4:12]):
    print(f'  step {i + 4:2d}: new {a:6.3f}  pack {b:6.3f}  reduce {c_:6.3f}  | ' + '  '.join(f'{k} {v:.3f}' for k, v in d.items()))
gaps = [a.elapsed_time(b) for a, b in gap_pairs[4:]]
print(f'  GPU idle between reduce end and the next mover begin: median {sorted(gaps)[len(gaps) // 2]:.3f} ms, min {min(gaps):.3f}, max {max(gaps):.3f}')
