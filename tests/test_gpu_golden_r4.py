"""HIP path vs tests/golden/r4.npz (the reference's own outputs, oracle/gen_golden.py --round4): scatter_* on INTEGER
tensors (reduce.py:6-23 hand any dtype to torch.index_reduce / index_add) — bit-exact, through the C ABI — and, from
ADVICE r3, moves / gathers / scatters through payloads whose base pointer is only 8-byte aligned."""
import numpy as np
import pytest
import torch

import torchrua_amd as ta
from gpu_util import DEV, assert_same_seq, dev_seq
from helpers import cases, golden, orc, to_np, to_torch

pytestmark = pytest.mark.gpu
OPS = ('max', 'min', 'sum', 'mean', 'prod')


@pytest.mark.parametrize('case', cases('scatter_int.'))
def test_integer_scatter_is_bit_exact(case):
    f = golden()[case]
    ten, idx, src = to_torch(f['tensor'], DEV), to_torch(f['index'], DEV), to_torch(f['source'], DEV)
    for name in OPS:
        for inc in (0, 1):
            got = getattr(ta, f'scatter_{name}')(ten, idx, src, include_self=bool(inc))
            want = f[f'scatter_{name}.{inc}']
            assert got.dtype == ten.dtype and tuple(got.shape) == want.shape
            np.testing.assert_array_equal(to_np(got), want, err_msg=f'{case} scatter_{name} include_self={inc}')
    assert to_np(ten).tobytes() == f['tensor'].tobytes()                       # the target is not written
    if 'last.tensor' in f:
        tt, ss = to_torch(f['last.tensor'], DEV), to_torch(f['last.source'], DEV)
        for name in OPS:
            for inc in (0, 1):
                got = getattr(ta, f'scatter_{name}')(tt, idx, ss, include_self=bool(inc), dim=-1)
                np.testing.assert_array_equal(to_np(got), f[f'last.scatter_{name}.{inc}'], err_msg=f'{case} last {name} {inc}')


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8])
@pytest.mark.parametrize('H', [1, 6, 16, 64, 200])
def test_integer_scatter_against_the_oracle(dtype, H):
    """Seeded mid-size inputs (many buckets, fan-in 0 .. a few hundred, int32 index, non-contiguous target) vs the
    oracle's restatement of ATen's integer index_reduce."""
    g = torch.Generator().manual_seed(1000 + H)
    S, Mn = 777, 20011
    info = torch.iinfo(dtype)
    lo, hi = max(info.min, -10 ** 6), min(info.max, 10 ** 6)
    idx = (torch.rand(Mn, generator=g) ** 3 * S).long().clamp_(0, S - 1)          # skewed: a few heavy buckets, empty ones
    ten = torch.randint(lo, hi + 1, (S, H), generator=g).to(dtype)
    src = torch.randint(lo, hi + 1, (Mn, H), generator=g).to(dtype)
    ten_d = ten.to(DEV).t().contiguous().t()                                      # column-major target
    for name in OPS:
        for inc in (False, True):
            got = getattr(ta, f'scatter_{name}')(ten_d, idx.to(DEV).int(), src.to(DEV), include_self=inc)
            want = getattr(orc, f'scatter_{name}')(ten.numpy(), idx.numpy(), src.numpy(), include_self=inc)
            np.testing.assert_array_equal(to_np(got), want, err_msg=f'{dtype} H={H} {name} {inc}')


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8])
def test_integer_scatter_against_the_calls_the_reference_makes(dtype):
    """reduce.py:6-23 ARE torch.index_reduce / torch.index_add: the same calls on the CPU of this box, on the same inputs
    (wrapping sums and products, counts beyond the type's range for int8 / uint8, 3-d targets along dim 1)."""
    import warnings
    g = torch.Generator().manual_seed(31)
    info = torch.iinfo(dtype)
    for S, Mn, shape_t, dim in ((5, 1500, (5, 6), 0), (64, 3000, (3, 64, 5), 1), (1, 40, (1,), 0)):
        idx = torch.randint(0, S, (Mn,), generator=g)
        shape_s = tuple(Mn if d == dim else n for d, n in enumerate(shape_t))
        ten = torch.randint(max(info.min, -7), min(info.max, 7) + 1, shape_t, generator=g).to(dtype)
        src = torch.randint(max(info.min, -7), min(info.max, 7) + 1, shape_s, generator=g).to(dtype)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for name, red in (('max', 'amax'), ('min', 'amin'), ('mean', 'mean'), ('prod', 'prod')):
                for inc in (False, True):
                    want = torch.index_reduce(ten, dim, idx, src, red, include_self=inc)
                    got = getattr(ta, f'scatter_{name}')(ten.to(DEV), idx.to(DEV), src.to(DEV), include_self=inc, dim=dim)
                    assert torch.equal(got.cpu(), want), f'{dtype} scatter_{name} include_self={inc} dim={dim}'
            for inc in (False, True):
                want = torch.index_add(ten if inc else torch.zeros_like(ten), dim, idx, src)
                got = ta.scatter_sum(ten.to(DEV), idx.to(DEV), src.to(DEV), include_self=inc, dim=dim)
                assert torch.equal(got.cpu(), want), f'{dtype} scatter_sum include_self={inc} dim={dim}'


def test_integer_scatter_edge_shapes():
    """No entries at all, one destination, empty rows, a 3-d target: what torch's own calls return on the CPU."""
    import warnings
    for shape_t, Mn in (((4, 3), 0), ((1, 5), 17), ((6, 0), 9), ((3, 2, 2), 8), ((5,), 0)):
        g = torch.Generator().manual_seed(len(shape_t) * 100 + Mn)
        S = shape_t[0]
        idx = torch.randint(0, S, (Mn,), generator=g)
        ten = torch.randint(-9, 10, shape_t, generator=g)
        src = torch.randint(-9, 10, (Mn,) + shape_t[1:], generator=g)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for name, red in (('max', 'amax'), ('min', 'amin'), ('mean', 'mean'), ('prod', 'prod')):
                for inc in (False, True):
                    want = torch.index_reduce(ten, 0, idx, src, red, include_self=inc)
                    got = getattr(ta, f'scatter_{name}')(ten.to(DEV), idx.to(DEV), src.to(DEV), include_self=inc)
                    assert got.shape == want.shape and torch.equal(got.cpu(), want), (shape_t, Mn, name, inc)
            for inc in (False, True):
                want = torch.index_add(ten if inc else torch.zeros_like(ten), 0, idx, src)
                got = ta.scatter_sum(ten.to(DEV), idx.to(DEV), src.to(DEV), include_self=inc)
                assert got.shape == want.shape and torch.equal(got.cpu(), want), (shape_t, Mn, 'sum', inc)


def test_integer_scatter_counts_tokens_at_scale():
    """The ordinary use: tokens per bucket = scatter_sum of ones on a long tensor; 4 M entries."""
    g = torch.Generator().manual_seed(7)
    S, Mn = 65536, 1 << 22
    idx = torch.randint(0, S, (Mn,), generator=g).to(DEV)
    got = ta.scatter_sum(torch.zeros(S, dtype=torch.long, device=DEV), idx, torch.ones(Mn, dtype=torch.long, device=DEV))
    assert torch.equal(got, torch.bincount(idx, minlength=S))
    wide = ta.scatter_max(torch.zeros(S, 4, dtype=torch.int32, device=DEV), idx,
                          torch.arange(Mn, device=DEV, dtype=torch.int32)[:, None].expand(Mn, 4).contiguous())
    last = torch.zeros(S, dtype=torch.long, device=DEV).scatter_reduce_(0, idx, torch.arange(Mn, device=DEV), 'amax')
    assert torch.equal(wide[:, 0].long(), last) and torch.equal(wide[:, 3].long(), last)


@pytest.mark.parametrize('dtype', [torch.int64, torch.int8, torch.uint8])
@pytest.mark.parametrize('H', [1, 5, 32, 130])
def test_integer_scatter_long_buckets(dtype, H):
    """Buckets longer than the reducer's part (1 024 rows below a million entries) are cut by position
    (rua_reduce_int.hip): neighbouring long buckets, one of exactly the part size (not long), one that starts on a cut,
    empty ones in between, everything in one bucket — against torch's own calls on the CPU."""
    import warnings
    info = torch.iinfo(dtype)
    for counts in ([3077, 0, 1, 1025, 1024, 7, 2048, 1, 1030, 0], [1024, 2049, 1025], [0, 0, 5000], [4099], [1] * 50 + [4050]):
        S = len(counts)
        g = torch.Generator().manual_seed(S * 1000 + H)
        idx = torch.repeat_interleave(torch.arange(S), torch.tensor(counts))
        idx = idx[torch.randperm(idx.numel(), generator=g)]
        Mn = idx.numel()
        ten = torch.randint(max(info.min, -3), min(info.max, 3) + 1, (S, H), generator=g).to(dtype)
        src = torch.randint(max(info.min, -3), min(info.max, 3) + 1, (Mn, H), generator=g).to(dtype)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for name, red in (('max', 'amax'), ('min', 'amin'), ('mean', 'mean'), ('prod', 'prod')):
                for inc in (False, True):
                    want = torch.index_reduce(ten, 0, idx, src, red, include_self=inc)
                    got = getattr(ta, f'scatter_{name}')(ten.to(DEV), idx.to(DEV), src.to(DEV), include_self=inc)
                    assert torch.equal(got.cpu(), want), f'{dtype} H={H} {counts[:4]} scatter_{name} include_self={inc}'
            for inc in (False, True):
                want = torch.index_add(ten if inc else torch.zeros_like(ten), 0, idx, src)
                got = ta.scatter_sum(ten.to(DEV), idx.to(DEV), src.to(DEV), include_self=inc)
                assert torch.equal(got.cpu(), want), f'{dtype} H={H} {counts[:4]} scatter_sum include_self={inc}'


def test_integer_scatter_skewed_histogram_at_scale():
    """A third of 4 M tokens in ONE bucket, a Zipf tail over the rest: counts == bincount, and the maximum row index per
    bucket (a 4-column int32 payload) == scatter_reduce's."""
    g = torch.Generator().manual_seed(11)
    S, Mn = 50000, 1 << 22
    u = torch.rand(Mn, generator=g)
    idx = torch.where(u < 0.33, torch.full((Mn,), 17), (torch.rand(Mn, generator=g) ** 4 * S).long().clamp_(0, S - 1)).to(DEV)
    got = ta.scatter_sum(torch.zeros(S, dtype=torch.long, device=DEV), idx, torch.ones(Mn, dtype=torch.long, device=DEV))
    assert torch.equal(got, torch.bincount(idx, minlength=S))
    wide = ta.scatter_max(torch.full((S, 4), -1, dtype=torch.int32, device=DEV), idx,
                          torch.arange(Mn, device=DEV, dtype=torch.int32)[:, None].expand(Mn, 4).contiguous(), include_self=True)
    last = torch.full((S,), -1, dtype=torch.long, device=DEV).scatter_reduce_(0, idx, torch.arange(Mn, device=DEV), 'amax')
    assert torch.equal(wide[:, 0].long(), last) and torch.equal(wide[:, 3].long(), last)


def test_integer_scatter_rejects_what_it_cannot_do():
    ten = torch.zeros(4, 2, dtype=torch.long, device=DEV)
    idx = torch.tensor([0, 1], device=DEV)
    src = torch.ones(2, 2, dtype=torch.long, device=DEV)
    assert ta.scatter_logsumexp(ten, idx, src).dtype == torch.float32      # [r5] answers like the reference (test_gpu_golden_r5.py)
    with pytest.raises(ta.RuaError):
        ta.scatter_logsumexp(ten.to(torch.uint8), idx, src.to(torch.uint8))   # the reference's own differences wrap: inf
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(ten, idx, src.int())                       # dtype mismatch, as torch
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(ten.bool(), idx, src.bool())


# ------------------------------------------------------------------ 8-byte-aligned base pointers
@pytest.mark.parametrize('row_bytes', [32, 64, 1024, 24, 40, 1000])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_moves_through_a_payload_at_an_8_byte_offset(row_bytes, dtype):
    """`flat[off:].view(N, H)` with off * itemsize = 8 (mod 16) is contiguous, so it reaches the mover as it is: rows
    of a multiple of 16 bytes then arrive with vec == 8 from the POINTER, not from the row size, and must not be taken
    for rows that end in an 8-byte tail (they lost their last 8 bytes).  Every layout, gather and scatter, vs the oracle."""
    es = torch.empty(0, dtype=dtype).element_size()
    H = row_bytes // es
    lens = np.array([3, 1, 7, 2, 5, 0, 4, 9, 1, 6, 2, 8, 3, 3, 1, 5, 7, 2, 4], dtype=np.int64)
    lens = np.concatenate([lens, lens[::-1], lens])                        # 57 sequences (ties: beyond 16)
    N = int(lens.sum())
    g = torch.Generator().manual_seed(row_bytes)
    off = 8 // es
    flat = torch.randn(N * H + off + 16, generator=g).to(dtype).to(DEV)
    data = flat[off:off + N * H].view(N, H)
    assert data.is_contiguous() and data.data_ptr() % 16 == 8
    c = ta.C(data, torch.from_numpy(lens).to(DEV))
    oc = orc.C(to_np(data), lens)
    bf = dtype == torch.bfloat16
    fill = 0.0
    # gathers out of the misaligned storage
    p = c.pack()
    op = orc.to_pack(oc, sorted_indices=to_np(p.sorted_indices))
    assert_same_seq(p, op, 'pack')
    assert_same_seq(c.left(fill), orc.to_left(oc, fill), 'left')
    assert_same_seq(c.right(fill), orc.to_right(oc, fill), 'right')
    assert_same_seq(c.roll(2), orc.roll(oc, 2), 'roll')
    # and INTO misaligned storage: a padded container living at the odd offset, then out again
    l0 = c.left(fill)
    B, T = l0.data.shape[:2]
    lflat = torch.empty(B * T * H + off + 16, dtype=dtype, device=DEV)
    ldata = lflat[off:off + B * T * H].view(B, T, H)
    ldata.copy_(l0.data)
    lm = ta.L(ldata, l0.token_sizes)
    assert ldata.data_ptr() % 16 == 8
    assert_same_seq(lm.cat(), oc, 'left(misaligned).cat')
    assert_same_seq(lm.pack(), op, 'left(misaligned).pack')
    assert_same_seq(lm.right(fill), orc.to_right(oc, fill), 'left(misaligned).right')
    # setitem (scatter mode) into the misaligned storage
    bp = torch.tensor([0, 2, 2, 7], device=DEV)
    tp = torch.tensor([1, 0, 6, 8], device=DEV)
    val = torch.randn(4, H, generator=g).to(dtype).to(DEV)
    c[bp, tp] = val
    oc2 = orc.C(to_np(data), lens)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    rows = offs[to_np(bp)] + to_np(tp)
    assert np.array_equal(to_np(data)[rows], to_np(val))
    assert_same_seq(c.pack(), orc.to_pack(oc2, sorted_indices=to_np(p.sorted_indices)), 'pack after setitem')
    # reductions read the same storage
    if not bf:
        got = ta.segment_sum(data, c.token_sizes)
        want = orc.segment_sum(to_np(data), lens)
        np.testing.assert_allclose(to_np(got), want, rtol=1e-5, atol=1e-5)       # fp32 sums: north_star's 1e-5
