"""Medium-size randomized parity vs the CPU oracle: big enough to cross every internal boundary
(multi-block scans: B > 2048; many mover tiles; sequences longer than the reducer's 64-row table block;
rows wider than one wave instruction), small enough for the oracle to finish in seconds."""
import numpy as np
import pytest
import torch

import torchrua_amd as ta
from gpu_util import DEV, assert_same_seq, dev_seq, host_sort
from helpers import orc, to_np

pytestmark = pytest.mark.gpu

SHAPES = [  # B, lo, hi, hidden, dtype
    (5000, 1, 40, (24,), torch.float32),
    (300, 1, 700, (136,), torch.bfloat16),     # 272-byte rows, T > 512
    (2049, 1, 9, (1,), torch.float32),          # scan tile boundary, 4-byte rows
    (70, 50, 400, (640,), torch.float32),       # 2 560-byte rows: 3 column chunks per row
    (120, 1, 90, (500,), torch.bfloat16),       # 1 000-byte rows (8 mod 16): 16-byte lanes + an 8-byte tail; 8-byte-lane reducers, one wave per row
    (90, 1, 80, (250,), torch.float32),         # the same width in fp32
    (150, 1, 60, (13,), torch.bfloat16),        # 26-byte rows: 2-byte lanes
    (40, 1, 50, (1004,), torch.bfloat16),       # 2 008-byte rows (8 mod 16, two column chunks)
    # [r5, late] rows of whole 16-byte vectors that are not a multiple of a 128-byte line, above 1 KiB: the LDS-staged
    # span kernel (eight rows of 2 000 bytes per tile; two of 6 000; 5 000-byte rows are 8 mod 16 again)
    (60, 1, 70, (1000,), torch.bfloat16),
    (50, 1, 40, (260,), torch.float32),         # 1 040-byte rows
    (30, 1, 30, (3000,), torch.bfloat16),       # 6 000-byte rows: two per tile
    (30, 1, 30, (2500,), torch.bfloat16),       # 5 000-byte rows (8 mod 16) beyond 4 KiB
    # [r5] narrow rows at sizes that reach the round-5 kernels (B >= 4096 for the segmented memcpy between batch-major
    # layouts, enough live cells for the 32 x 64 .. 128 x 128 transposing tiles, the full-grid pads out of a
    # PackedSequence, the step-by-step roll): 1-D payloads of 8 / 4 / 2 / 1-byte elements, and 12 .. 64-byte rows
    (6000, 1, 40, (), torch.int64),
    (6000, 1, 40, (), torch.float32),
    (6000, 1, 40, (), torch.int16),
    (6000, 1, 40, (), torch.uint8),
    (6000, 1, 40, (2,), torch.int64),
    (6000, 1, 40, (3,), torch.float32),
    (5000, 1, 40, (16,), torch.bfloat16),
    (4500, 1, 40, (16,), torch.float32),
]


def _inputs(shape):
    B, lo, hi, hidden, dtype = shape
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn((int(lens.sum()),) + hidden, generator=g)
    if not dtype.is_floating_point:
        data = data * 1000 if dtype != torch.uint8 else data.abs() * 60
    return lens, data.to(dtype)


def _id(s):
    return f'B{s[0]}-T{s[2]}-H{s[3][0] if s[3] else "vec"}-{str(s[4])[6:]}'


@pytest.mark.parametrize('shape', SHAPES, ids=[_id(s) for s in SHAPES])
def test_casts_selects_vs_oracle(shape):
    lens, data = _inputs(shape)
    bf = data.dtype == torch.bfloat16
    srt = host_sort(lens)
    oc = orc.C(to_np(data), lens.numpy())
    fv = -3.0 if data.dtype.is_floating_point else (3 if data.dtype == torch.uint8 else -3)
    fill = np.array([0xC040], dtype=np.uint16) if bf else fv          # bf16 bits of -3.0
    osq = {'C': oc, 'L': orc.to_left(oc, fill), 'P': orc.to_pack(oc, srt), 'R': orc.to_right(oc, fill)}
    dsq = {k: dev_seq(v, bf16=bf) for k, v in osq.items()}
    for k, z in dsq.items():
        for dst in 'CLPR':
            out = {'C': z.cat, 'P': z.pack, 'L': lambda: z.left(fv), 'R': lambda: z.right(fv)}[dst]()
            assert_same_seq(out, orc.to_kind(osq[k], dst, fill, srt), f'{k}->{dst}')
        bp, tp = z.ptr()
        obp, otp = orc.ptr(osq[k])
        assert np.array_equal(to_np(bp), obp) and np.array_equal(to_np(tp), otp), f'ptr {k}'
        assert np.array_equal(to_np(z.idx().data), orc.idx(osq[k]).data), f'idx {k}'
        assert np.array_equal(to_np(ta.get_mask(z)), orc.get_mask(osq[k])), f'mask {k}'
        for s in (1, -7, 123, 2, -3):
            assert_same_seq(z.roll(s), orc.roll(osq[k], s, srt), f'roll {k} {s}')
        assert_same_seq(z.rev(), orc.rev(osq[k], srt), f'rev {k}')
        nz = lens.numpy() > 0      # (`last` of an EMPTY sequence: the reference reads the row in front of it, we write zeros — DESIGN §5)
        assert np.array_equal(to_np(z.last())[nz], orc.last(osq[k])[nz]), f'last {k}'
        m = max(1, int(lens.min()))
        a, b = (m - 1) // 2, (m - 1) - (m - 1) // 2
        t = z.trunc((a, b))
        assert_same_seq(t._replace(data=t.data.contiguous()), orc.trunc(osq[k], (a, b)), f'trunc {k}')


@pytest.mark.parametrize('dtype,hidden', [(torch.int64, ()), (torch.float32, ()), (torch.uint8, ()), (torch.float32, (3,))])
def test_narrow_rows_with_empty_sequences(dtype, hidden):
    """The round-5 narrow-row kernels over a batch that holds EMPTY sequences (the reference — and so the oracle — cannot
    take those through a PackedSequence): expectations built with plain torch on the host."""
    g = torch.Generator().manual_seed(77)
    B = 5000
    lens = torch.randint(0, 71, (B,), generator=g)
    lens[::97] = 0
    n, T = int(lens.sum()), int(lens.max())
    data = (torch.randn((n,) + hidden, generator=g) * 50).abs().to(dtype)
    seqs = list(torch.split(data, lens.tolist()))
    fill = 7
    left = torch.full((B, T) + hidden, fill, dtype=dtype)
    right = torch.full((B, T) + hidden, fill, dtype=dtype)
    for b, x in enumerate(seqs):
        left[b, :x.size(0)] = x
        right[b, T - x.size(0):] = x
    rolled = torch.cat([torch.roll(x, 2, 0) for x in seqs])
    c = ta.C(data.to(DEV), lens.to(DEV))
    p = c.pack()
    for name, got, want in (('C.left', c.left(fill).data, left), ('C.right', c.right(fill).data, right),
                            ('P.left', p.left(fill).data, left), ('P.right', p.right(fill).data, right),
                            ('P.cat', p.cat().data, data), ('L.cat', c.left(fill).cat().data, data),
                            ('R.cat', c.right(fill).cat().data, data), ('R.left', c.right(fill).left(fill).data, left),
                            ('L.right', c.left(fill).right(fill).data, right), ('L.pack.cat', c.left(fill).pack().cat().data, data),
                            ('C.roll', c.roll(2).data, rolled), ('P.roll', p.roll(2).cat().data, rolled),
                            ('R.pack', c.right(fill).pack().data, p.data.cpu())):
        assert torch.equal(got.cpu(), want), name
    assert torch.equal(p.cat().token_sizes.cpu(), lens)


RED_SHAPES = SHAPES[:2] + SHAPES[3:8]        # (the floating-point ones)


@pytest.mark.parametrize('shape', RED_SHAPES, ids=[f'B{s[0]}-H{s[3][0]}-{str(s[4])[6:]}' for s in RED_SHAPES])
def test_reductions_vs_oracle(shape):
    lens, data = _inputs(shape)
    data = (data * 0.25).to(data.dtype)
    f32 = data.float().numpy()
    c = ta.C(data.to(DEV), lens.to(DEV))
    p = c.pack()
    ulp = 2.0 ** -8 if data.dtype == torch.bfloat16 else 0.0
    for name in ('sum', 'mean', 'max', 'min', 'logsumexp'):
        ref = getattr(orc, f'segment_{name}')(f32, lens.numpy())
        for what, got in (('segment', getattr(ta, f'segment_{name}')(c.data, c.token_sizes)),
                          ('reduce(P)', getattr(ta, f'reduce_{name}')(p)),
                          ('reduce(L)', getattr(ta, f'reduce_{name}')(c.left())),
                          ('fused', ta.pack_reduce(c, name, fused=True)[1])):
            # 1e-5 on the fp32 accumulation (scaled by the magnitude that was summed) + output rounding
            scale = np.abs(f32).max() * (lens.max().item() if name == 'sum' else 1)
            np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=1e-5 + ulp, atol=1e-5 * scale + 1e-6,
                                       err_msg=f'{what} {name}')
    # scatter: rows shuffled, destinations = sequence ids
    index = torch.repeat_interleave(torch.arange(lens.numel()), lens)
    perm = torch.randperm(index.numel(), generator=torch.Generator().manual_seed(1))
    if data.dtype == torch.float32:
        ten = torch.randn((lens.numel(),) + tuple(data.shape[1:]))
        for name in ('sum', 'max', 'mean'):
            ref = getattr(orc, f'scatter_{name}')(ten.numpy(), index[perm].numpy(), f32[perm.numpy()], include_self=True)
            got = getattr(ta, f'scatter_{name}')(ten.to(DEV), index[perm].to(DEV), data[perm].to(DEV), include_self=True)
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-4, err_msg=f'scatter {name}')


def test_bf16_outputs_are_the_rounded_fp32_oracle():
    """accumulate in fp32, round ONCE: the bf16 result equals oracle.bfloat16() up to 1 ulp
    (the reference's own bf16 path accumulates in bf16 and is ~1.5 % off, so it cannot be the yardstick)."""
    lens, data = _inputs((300, 1, 700, (136,), torch.bfloat16))
    c = ta.C(data.to(DEV), lens.to(DEV))
    p = c.pack()
    f32 = data.float().numpy()
    for name in ('sum', 'mean'):
        ref = torch.from_numpy(getattr(orc, f'segment_{name}')(f32, lens.numpy())).bfloat16()
        for got in (getattr(ta, f'segment_{name}')(c.data, c.token_sizes), getattr(ta, f'reduce_{name}')(p)):
            a = got.cpu().view(torch.int16).to(torch.int32)
            b = ref.view(torch.int16).to(torch.int32)
            # adjacent bf16 values differ by 1 in their bit patterns (same sign); +-0 are 0x0000 / 0x8000
            ulps = (a - b).abs()
            same_sign = (a < 0) == (b < 0)
            assert bool(((ulps <= 1) & same_sign | (ref.float().abs() < 1e-3)).all()), name
            assert float((ulps == 0).float().mean()) > 0.97     # and almost always identical
