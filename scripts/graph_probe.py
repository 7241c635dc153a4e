"""Developer probe: which ops survive HIP-graph capture."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

DEV = torch.device('cuda:0')
g = torch.Generator().manual_seed(11)
lens = torch.randint(1, 30, (64,), generator=g)
n = int(lens.sum())
static = torch.randn(n, 32, generator=g).to(DEV)
c = ta.with_host_sizes(static, lens)
p0 = c.pack()
OPS = {
    'reduce_sum(p0)': lambda: ta.reduce_sum(p0),
    'reduce_max(p0)': lambda: ta.reduce_max(p0),
    'p0.roll': lambda: p0.roll(1).data,
    'p0.last': lambda: p0.last(),
    'c.left': lambda: c.left(-1.0).data,
    'p0.cat': lambda: p0.cat().data,
    'segment_sum': lambda: ta.segment_sum(c.data, c.token_sizes),
    'segment_max': lambda: ta.segment_max(c.data, c.token_sizes),
    'c.pack': lambda: c.pack().data,
}
for name, fn in OPS.items():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            out = fn()
        graph.replay()
        torch.cuda.synchronize()
        print(f'{name:18s} ok', bool(torch.equal(out, fn())), flush=True)
    except Exception as e:  # noqa: BLE001
        print(f'{name:18s} FAILED {type(e).__name__}: {str(e).splitlines()[0]}', flush=True)
        torch.cuda.synchronize()
