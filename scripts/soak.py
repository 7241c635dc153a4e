"""Randomized soak (developer tool; uses the CPU oracle as the checker like tests/ do): random batch shapes
large enough to cross tile / scan / table-block boundaries, every cast, roll, rev, last, reductions, fused
pack_reduce, against oracle/rua_oracle.  Usage: python scripts/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torchrua_amd as ta  # noqa: E402
from gpu_util import DEV, assert_same_seq, dev_seq, host_sort  # noqa: E402
from helpers import orc, to_np  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.RandomState(seed)
print('seed', seed)
t_end = time.time() + budget
n = 0
last_report = time.time()
while time.time() < t_end:
    family = rng.choice(['mixed', 'long', 'many'], p=[0.7, 0.15, 0.15])
    if family == 'mixed':
        B = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 257, 1000, 2049, 3000]))
        hi = int(rng.choice([1, 2, 5, 17, 63, 64, 65, 130, 300, 700]))
        H = int(rng.choice([1, 2, 3, 4, 8, 16, 24, 31, 32, 64, 96, 100, 128, 136, 250, 256, 260, 500, 520, 1000, 1024, 1080, 1250]))
    elif family == 'long':       # few sequences, some far beyond the reducer's part size: split / tail / combine
        B = int(rng.choice([1, 3, 11, 40]))
        hi = int(rng.choice([3000, 9000, 30000]))
        H = int(rng.choice([3, 8, 64, 130, 512]))
    else:                        # so many short sequences that narrow rows take the adjacent-rank kernels
        B = int(rng.choice([40000, 140000]))
        hi = int(rng.choice([1, 2, 4, 9, 30, 150]))      # (150: four sequences per wave at rows of <= 32 bytes)
        H = int(rng.choice([1, 2, 4, 8, 16, 24, 64, 128]))
    lo = int(rng.randint(1, hi + 1))
    dtype = [torch.float32, torch.bfloat16, torch.float16, torch.float64, torch.int64][int(rng.randint(0, 5))]
    if B * hi * H > 6e7:
        continue
    lens = torch.from_numpy(rng.randint(lo, hi + 1, size=B).astype(np.int64))
    if family == 'long':
        lens[rng.randint(0, B)] = hi          # at least one really long sequence next to random ones
        lens[rng.randint(0, B)] = 1
    if family == 'many' and rng.randint(0, 2):
        for _ in range(5):                    # a few outliers: the waves that hold them walk their sequences one by one
            lens[rng.randint(0, B)] = 40 * hi + 100
    empties = family != 'long' and B >= 7 and rng.randint(0, 10) < 3
    if empties:                               # [r5] every tenth sequence empty (max / min / logsumexp: the global `initial`)
        lens[torch.from_numpy(rng.randint(0, 10, size=B)) == 0] = 0
        if int(lens.sum()) == 0:
            lens[0] = hi
    g = torch.Generator().manual_seed(int(rng.randint(0, 2 ** 31)))
    N = int(lens.sum())
    data = torch.randint(-99, 99, (N, H), generator=g) if dtype == torch.int64 else (torch.randn(N, H, generator=g) * 0.5).to(dtype)
    bf = dtype == torch.bfloat16
    tag = f'B={B} len=[{lo},{hi}] H={H} {dtype}' + (' +empties' if empties else '')
    try:
        srt = host_sort(lens)
        oc = orc.C(to_np(data), lens.numpy())
        if empties:
            # (the reference — and so its restatement — raises on a PackedSequence that holds zero-length sequences: the
            # casts are soaked without them; the reductions get the library's own PackedSequence)
            osq = {'C': oc, 'L': orc.to_left(oc, 0), 'R': orc.to_right(oc, 0)}
            dsq = {k: dev_seq(v, bf16=bf) for k, v in osq.items()}
            dsq['P'] = dsq['C'].pack()
        else:
            osq = {'C': oc, 'L': orc.to_left(oc, 0), 'P': orc.to_pack(oc, srt), 'R': orc.to_right(oc, 0)}
            dsq = {k: dev_seq(v, bf16=bf) for k, v in osq.items()}
        for k, z in ([] if empties else dsq.items()):
            for dst in 'CLPR':
                out = {'C': z.cat, 'P': z.pack, 'L': z.left, 'R': z.right}[dst]()
                assert_same_seq(out, orc.to_kind(osq[k], dst, 0, srt), f'{k}->{dst}')
            s = int(rng.randint(-hi - 2, hi + 3))
            assert_same_seq(z.roll(s), orc.roll(osq[k], s, srt), f'roll {k} {s}')
            assert_same_seq(z.rev(), orc.rev(osq[k], srt), f'rev {k}')
            nz = lens.numpy() > 0      # (`last` of an EMPTY sequence: the reference reads the row in front of it, we write zeros)
            assert np.array_equal(to_np(z.last())[nz], orc.last(osq[k])[nz]), f'last {k}'
            if N <= 3_000_000:          # the integer kernels (ptr / idx / masks), bit-exact
                bp, tp = z.ptr()
                obp, otp = orc.ptr(osq[k])
                assert np.array_equal(to_np(bp), obp) and np.array_equal(to_np(tp), otp), f'ptr {k}'
                assert np.array_equal(to_np(z.idx().data), orc.idx(osq[k]).data), f'idx {k}'
                if B * int(lens.max()) <= 4_000_000:
                    assert np.array_equal(to_np(z.bmask()), orc.mask(osq[k], False, True, np.bool_)), f'bmask {k}'
                    assert np.array_equal(to_np(ta.get_mask(z)), orc.get_mask(osq[k])), f'get_mask {k}'
        if dtype != torch.int64:
            f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
            ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
            for name in ('sum', 'mean', 'max', 'min', 'logsumexp'):
                ref = getattr(orc, f'segment_{name}')(f, lens.numpy()).astype(np.float64)
                scale = float(np.abs(f).max()) * (hi if name == 'sum' else 1)
                outs = [getattr(ta, f'reduce_{name}')(dsq[k]) for k in 'CLPR']
                # lengths the host knows: short sequences side by side in a wave (RUA_OP_SHORT_SEQS), exact split decisions
                outs.append(getattr(ta, f'reduce_{name}')(ta.with_host_sizes(dsq['C'].data, lens)))
                if H % (16 // data.element_size()) == 0:
                    outs.append(ta.pack_reduce(dsq['C'], name)[1])
                for o in outs:
                    np.testing.assert_allclose(o.double().cpu().numpy(), ref, rtol=2e-5 + ulp, atol=2e-5 * scale + ulp + 1e-6,
                                               err_msg=name)
            if dtype in (torch.float32, torch.float64):
                # fused backward of every layout vs autograd through torch.segment_reduce on the CPU (what the
                # reference calls, reduce.py:34-53); ties of max/min share the gradient in both
                name = ['sum', 'mean', 'max', 'min'][int(rng.randint(0, 4))]
                tied = (torch.randint(0, 3, (N, H), generator=g)).to(dtype)
                # cotangents of both signs: torch's segment_reduce backward shares a gradient among tied extrema only
                # when it is > 0 (its kernel tests `grad_input > 0` to find them) — RUA_BWD_TIES_POSITIVE does the same
                cot = torch.randn(B, H, generator=g).to(dtype)
                r = tied.clone().requires_grad_(True)
                torch.segment_reduce(r, name, lengths=lens, unsafe=True).backward(cot)
                for k in 'CLPR':
                    x = tied.clone().to(DEV).requires_grad_(True)
                    z = {'C': lambda c_: c_, 'L': lambda c_: c_.left(), 'P': lambda c_: c_.pack(),
                         'R': lambda c_: c_.right()}[k](ta.with_host_sizes(x, lens))
                    getattr(ta, f'reduce_{name}')(z).backward(cot.to(DEV))
                    torch.testing.assert_close(x.grad.cpu(), r.grad, rtol=1e-5, atol=1e-6, msg=f'backward {name} {k}')
            if dtype == torch.float32 and N * H < 4e6:
                # scatter_* over shuffled rows (both include_self values) and gradients of a reduce / a cast chain
                index = torch.repeat_interleave(torch.arange(B), lens)
                perm = torch.randperm(N, generator=g)
                ten = torch.randn(B, H, generator=g)
                for name in ('sum', 'max', 'mean'):
                    for inc in (False, True):
                        ref = getattr(orc, f'scatter_{name}')(ten.numpy(), index[perm].numpy(), f[perm.numpy()], include_self=inc)
                        got = getattr(ta, f'scatter_{name}')(ten.to(DEV), index[perm].to(DEV), data[perm].to(DEV), include_self=inc)
                        # fp32 sums of up to `max len` terms in different orders: the error scales with the length
                        sscale = float(np.abs(f).max()) * (int(lens.max()) if name == 'sum' else 1)
                        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * sscale + 1e-4, err_msg=f'scatter {name}')
                x = data.clone().to(DEV).requires_grad_(True)
                c = ta.C(x, lens.to(DEV))
                out = ta.reduce_logsumexp(c.pack().roll(1).left().pack())
                cot = torch.randn(out.shape, generator=g).to(DEV)
                out.backward(cot)
                r = data.clone().to(DEV).requires_grad_(True)
                torch.stack([t_.logsumexp(0) for t_ in torch.split(r, lens.tolist())]).backward(cot)
                torch.testing.assert_close(x.grad, r.grad, rtol=1e-4, atol=1e-5)
        if dtype == torch.int64 and N * H < 4e6:
            # scatter_* on integer tensors (reduce.py:6-23), bit-exact against the oracle's restatement of ATen's fold
            idt = [torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8][int(rng.randint(0, 5))]
            info = torch.iinfo(idt)
            ten = torch.randint(max(info.min, -50), min(info.max, 50) + 1, (B, H), generator=g).to(idt)
            index = torch.repeat_interleave(torch.arange(B), lens)
            perm = torch.randperm(N, generator=g)
            src = data.clamp(info.min, info.max).to(idt)[perm]
            for name in ('sum', 'max', 'min', 'prod', 'mean'):
                for inc in (False, True):
                    ref = getattr(orc, f'scatter_{name}')(ten.numpy(), index[perm].numpy(), src.numpy(), include_self=inc)
                    got = getattr(ta, f'scatter_{name}')(ten.to(DEV), index[perm].to(DEV), src.to(DEV), include_self=inc)
                    assert np.array_equal(to_np(got), ref), f'integer scatter_{name} {idt} include_self={inc}'
        # the bucket builder itself against a stable sort, at destination counts on both sides of every path switch
        # (one LSD pass up to 512, the two-level MSD builder up to 262 144, LSD passes beyond), out-of-range entries too
        from torchrua_amd import _ops as O
        S2 = int(rng.choice([1, 2, 300, 512, 513, 4096, 65536, 70000, 262144, 262145, 300000]))
        M2 = int(rng.choice([1, 100, 4095, 4097, 8193, 50000, 400000]))
        idx2 = torch.randint(-1, S2 + 1, (M2,), generator=g)
        if rng.randint(0, 3) == 0:
            idx2[::3] = int(rng.randint(0, S2))            # a heavy destination
        idx2 = idx2.to(DEV)
        counts2, perm2 = O.index_buckets(idx2, S2)
        ok2 = (idx2 >= 0) & (idx2 < S2)
        order2 = torch.sort(torch.where(ok2, idx2, torch.full_like(idx2, S2)), stable=True)[1]
        n_ok2 = int(ok2.sum())
        assert torch.equal(perm2[:n_ok2], order2[:n_ok2]), f'index_buckets perm S={S2} M={M2}'
        assert torch.equal(counts2, torch.bincount(idx2[ok2], minlength=S2)), f'index_buckets counts S={S2} M={M2}'
    except Exception:
        print('FAILED at', tag)
        raise
    n += 1
    if time.time() - last_report > 30:
        print(f'  {n} configurations so far', flush=True)
        last_report = time.time()
print(f'soak ok: {n} random configurations')
