"""Developer probe: segment_max vs segment_sum kernel time on mid-size shapes (which library: RUA_LIB_PATH)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402

dev = torch.device('cuda:0')


class Timer:
    def __init__(self):
        self.open, self.ev = {}, []

    def __call__(self, name, begin):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if begin:
            self.open[name] = ev
        else:
            self.ev.append((self.open.pop(name), ev))

    def median(self):
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in self.ev)
        return ts[len(ts) // 2] * 1e3


def run(fn, iters=20):
    t = Timer()
    fn()
    torch.cuda.synchronize()
    O.set_kernel_hook(t)
    for _ in range(iters):
        fn()
    O.set_kernel_hook(None)
    return t.median()


print('lib', os.environ.get('RUA_LIB_PATH', 'default'))
for (B, lo, hi, H) in ((512, 8, 512, 512), (4096, 8, 512, 256), (8192, 8, 512, 512), (16384, 1, 64, 512), (65536, 8, 512, 512)):
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    row = [f'B={B:6d} H={H:4d}']
    for name in ('sum', 'max', 'logsumexp'):
        row.append(f'{name}(C) {run(lambda: getattr(ta, "segment_" + name)(c.data, c.token_sizes)):8.1f}')
        row.append(f'{name}(P) {run(lambda: getattr(ta, "reduce_" + name)(p)):8.1f}')
    print(' | '.join(row), flush=True)
    del data, c, p
