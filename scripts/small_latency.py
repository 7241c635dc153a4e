"""Developer probe: host-visible latency of single ops on tiny inputs (BASELINE cfg1: B=32, U(4,64), H=32, fp32)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
lens = torch.randint(4, 65, (32,), generator=g)
xs = [torch.randn(int(n), 32, device=dev) for n in lens]


def t(name, fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f'{name:34s} {(time.perf_counter() - t0) / n * 1e6:9.1f} us/call')


c = ta.C.new(xs)
p = c.pack()
l = c.left()
t('C.new(xs)  (torch.cat + sizes)', lambda: ta.C.new(xs))
t('C.new(xs).left()   [cfg1]', lambda: ta.C.new(xs).left())
t('c.left()', lambda: c.left())
t('c.pack()', lambda: c.pack())
t('p.cat()', lambda: p.cat())
t('p.roll(1)', lambda: p.roll(1))
t('p.last()', lambda: p.last())
t('reduce_sum(p)', lambda: ta.reduce_sum(p))
t('segment_max(c)', lambda: ta.segment_max(c.data, c.token_sizes))
t('c.ptr()', lambda: c.ptr())
from torch.nn.utils.rnn import pack_sequence, pad_sequence  # noqa: E402
t('torch pad_sequence(xs)', lambda: pad_sequence(xs, batch_first=True))
t('torch pack_sequence(xs)', lambda: pack_sequence(xs, enforce_sorted=False))


# the same ops replayed from one HIP graph (fixed lengths, new payload copied into the static input)
def chain():
    q = c.pack()
    return ta.reduce_sum(q), q.roll(1).data, q.last(), c.left().data


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    chain()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    outs = chain()
t('pack+reduce+roll+last+left eager', chain)
t('the same five ops, graph.replay()', graph.replay)
