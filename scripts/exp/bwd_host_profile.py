"""Developer probe: where the HOST time of a small fused-reduce backward goes — _ReduceBwd.forward called directly (so that
cProfile, which does not follow the autograd engine's device thread, sees it) over a CattedSequence and a PackedSequence of
4 096 short sequences; then the same through torch.autograd.grad, per call."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as L, _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
lens = torch.randint(1, 32, (4096,), generator=g)
data = torch.randn(int(lens.sum()), 64, device=dev)
c = ta.with_host_sizes(data, lens)
for zname, z in (('C', c), ('P', c.pack())):
    lay = describe(z)
    out = ta.reduce_sum(z)
    cot = torch.ones_like(out)
    fn = lambda: O._ReduceBwd.apply(cot, z.data, out, lay, L.SUM, None)      # noqa: E731
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    t_host = (time.perf_counter() - t0) / 2000 * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 2000 * 1e6
    print(f'{zname}: {t_host:.1f} us of host time per direct call ({t_all:.1f} us with the queue drained)')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
