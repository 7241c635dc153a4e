"""Developer probe: integer scatter_sum with and without the long-bucket split of rua_reduce_int.hip, by how skewed the
histogram is (17 M int64 ones -> S buckets)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as K  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402
from torchrua_amd import _ops as O  # noqa: E402

dev = torch.device('cuda:0')
Mn = 17_046_960


def med(fn, rounds=5):
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


g = torch.Generator().manual_seed(3)
for H in (1, 64):
    src = torch.ones(Mn, H, dtype=torch.long, device=dev)
    for name, S, idx in (('uniform, 65 536 buckets', 65536, torch.randint(0, 65536, (Mn,), generator=g)),
                         ('a third in one bucket, 65 536', 65536, torch.where(torch.rand(Mn, generator=g) < 0.33, torch.tensor(5), torch.randint(0, 65536, (Mn,), generator=g))),
                         ('10 buckets', 10, torch.randint(0, 10, (Mn,), generator=g)),
                         ('one bucket', 1, torch.zeros(Mn, dtype=torch.long))):
        idx = idx.to(dev)
        counts, perm = O.index_buckets(idx, S)
        lay = M.lay_cat(counts, S, Mn)
        out = torch.empty(S, H, dtype=torch.long, device=dev)
        split, ws = O.int_split_workspace(Mn, H, torch.long, dev)
        lib = K.load()

        def run(sp, w):
            K.check(lib.rua_segment_reduce(lay.ref(), K.ptr(perm), K.ptr(src), K.ptr(out), H, K.INT_DTYPES[torch.long], K.SUM, 0, 0,
                                           None, sp, K.ptr(w), None, K.stream_ptr(dev)), 'reduce')
        t1 = med(lambda: run(split, ws))
        ref = out.clone()
        t0 = med(lambda: run(0, None))
        assert torch.equal(ref, out)
        tt = med(lambda: torch.zeros(S, H, dtype=torch.long, device=dev).index_add_(0, idx, src), rounds=3)
        print(f'H={H:3d} {name:32s}: reducer alone {t0:8.3f} ms, with the split {t1:8.3f} ms (parts of {split} rows); torch index_add_ {tt:8.3f} ms', flush=True)
