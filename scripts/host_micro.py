"""Developer probe: microseconds of the host-side pieces of pack() at B = 4096."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _meta as M, _lib as K, core  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B = int(os.environ.get('RUA_PROBE_B', 4096))
lens = torch.randint(8, 513, (B,), generator=g)
data = torch.randn(int(lens.sum()), int(os.environ.get('RUA_PROBE_H', 1)), device=dev, dtype=torch.bfloat16)


def t(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f'{name:46s} {dt:8.1f} us')


def hs():
    with M.host_serial():
        pass


t('host_serial enter/exit', hs)
t('torch.sort(lens, descending) [default threads]', lambda: torch.sort(lens, descending=True))
def srt():
    with M.host_serial():
        torch.sort(lens, descending=True)
t('torch.sort under host_serial', srt)
idx = torch.sort(lens, descending=True)[1]
t('to_device_async(index)', lambda: M.to_device_async(idx, dev))
t('batch_sizes_from_host_lens', lambda: M.batch_sizes_from_host_lens(lens, 512))
t('with_host_sizes', lambda: ta.with_host_sizes(data, lens))
c = ta.with_host_sizes(data, lens)
t('max_len + total_len (fresh C)', lambda: (lambda cc: (M.max_len(cc.token_sizes), M.total_len(cc.token_sizes)))(ta.with_host_sizes(data, lens)))
t('torch.empty x4', lambda: [torch.empty(10, dtype=torch.long, device=dev) for _ in range(4)])
t('stream_ptr', lambda: K.stream_ptr(dev))
def full():
    cc = ta.with_host_sizes(data, lens)
    return cc.pack()
t('with_host_sizes + pack()  (all host work + launches)', full)
p = full()
t('reduce_sum(p) (fresh p each)', lambda: ta.reduce_sum(full()))
