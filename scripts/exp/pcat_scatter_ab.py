"""Developer A/B: P.cat at wide rows as a GATHER (destination-ordered tiles: 16 consecutive tokens of one sequence, read from 16 time
steps) against a SCATTER (source-ordered tiles: 16 consecutive rows of the PackedSequence, written to 16 sequences), and the pack
both ways, at the north-star shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as K  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402
from torchrua_amd.layout import describe  # noqa: E402

dev = torch.device('cuda:0')


def med(fn, rounds=9):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for B, H in ((65536, 512), (65536, 128), (16384, 512)):
    lens = torch.randint(8, 513, (B,), generator=torch.Generator().manual_seed(5))
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    cl, pl = describe(c), describe(p)
    out = torch.empty_like(data)
    line = f'B={B} H={H}:'
    for name, dst, src, x, want in (('P.cat', cl, pl, p.data, data), ('pack', pl, cl, data, p.data)):
        tg = med(lambda: O.launch_move(O.MovePlan(dst, src, data.shape), x, out=out))
        assert torch.equal(out, want), name + ' gather'
        out.zero_()
        # (scatter form: the FIRST layout enumerates the rows of the payload handed in, the second places them)
        ts = med(lambda: O.launch_move(O.MovePlan(src, dst, data.shape, flags=K.MOVE_SCATTER), x, out=out))
        assert torch.equal(out, want), name + ' scatter'
        line += f'  {name}: gather {tg:6.3f} ms, scatter {ts:6.3f} ms ({(tg / ts - 1) * 100:+5.1f} %)'
    print(line, flush=True)
    del data, c, p, out
    torch.cuda.empty_cache()
