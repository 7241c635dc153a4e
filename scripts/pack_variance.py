"""Developer probe: how much does the north-star pack kernel vary (a) step to step inside one process,
(b) with the address of its output buffer?  Times the C->P mover alone (HIP events), output buffer placed by hand."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
B, H = 65536, 512
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
data = torch.empty(N, H, device=dev, dtype=torch.bfloat16).normal_()
c = ta.with_host_sizes(data, lens)
p = c.pack()
cl, pl = describe(c), describe(p)
plan = O.MovePlan(pl, cl, data.shape)
nbytes = 2 * N * H * 2
pool = torch.empty(N * H * 2 + (64 << 20), dtype=torch.uint8, device=dev)     # room to slide the output by up to 64 MiB


def run(out, reps):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        O.launch_move(plan, data, out=out)
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ts]


print(f'src at {data.data_ptr():#x}, pool at {pool.data_ptr():#x}')
out0 = pool[:N * H * 2].view(torch.bfloat16).view(N, H)
run(out0, 3)
ts = run(out0, 40)
print('40 consecutive launches, same buffers (ms):', ' '.join(f'{t:.2f}' for t in ts))
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 8 << 20, 13 << 20, 32 << 20, 63 << 20):
    out = pool[off:off + N * H * 2].view(torch.bfloat16).view(N, H)
    ts = sorted(run(out, 7))
    print(f'output offset {off:>10d} B: median {ts[3]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}   {nbytes / ts[3] / 1e9:.2f} TB/s', flush=True)
