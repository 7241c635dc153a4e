"""Edge cases through the C ABI vs the CPU oracle: odd row widths (every vector width of the mover and
the scalar path of the reducer), rows wider than a wave instruction, multi-dim hidden, integer payloads,
single-token / single-sequence batches, zero-length segments, non-contiguous inputs, huge shifts."""
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import torchrua_amd as ta
from gpu_util import DEV, assert_same_seq, dev_seq, host_sort
from helpers import orc, to_np

pytestmark = pytest.mark.gpu


def _mk(lens, hidden, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = int(sum(lens))
    if dtype in (torch.long, torch.int32, torch.uint8, torch.int16):
        data = torch.randint(0, 100, (n,) + tuple(hidden), generator=g).to(dtype)
    else:
        data = torch.randn((n,) + tuple(hidden), generator=g).to(dtype)
    return data, torch.tensor(lens, dtype=torch.long)


WIDTHS = [((1,), torch.uint8), ((3,), torch.uint8), ((1,), torch.int16), ((3,), torch.float16), ((1,), torch.float32),
          ((33,), torch.float32), ((6,), torch.float32), ((1,), torch.long), ((2,), torch.long), ((5,), torch.long),
          ((300,), torch.float32), ((1024,), torch.bfloat16), ((1030,), torch.bfloat16), ((4, 3), torch.float32),
          ((2, 3, 5), torch.float64), ((), torch.float32), ((513,), torch.float16),
          ((70001,), torch.float32), ((65536,), torch.bfloat16)]      # 273-KiB odd rows, 128-KiB rows


@pytest.mark.parametrize('hidden,dtype', WIDTHS, ids=[f'{h}-{str(d)[6:]}' for h, d in WIDTHS])
def test_all_casts_and_selects_any_row_width(hidden, dtype):
    lens = [3, 1, 7, 2, 7, 4, 1, 5, 6, 2, 3, 3, 9, 1, 2, 8, 4, 4, 2]
    data, lt = _mk(lens, hidden, dtype)
    bf = dtype == torch.bfloat16
    srt = host_sort(lens)
    oc = orc.C(to_np(data), lt.numpy())
    fill = 0
    osq = {'C': oc, 'L': orc.to_left(oc, fill), 'P': orc.to_pack(oc, srt), 'R': orc.to_right(oc, fill)}
    dsq = {k: dev_seq(v, bf16=bf) for k, v in osq.items()}
    for k, z in dsq.items():
        for dst in 'CLPR':
            out = {'C': z.cat, 'P': z.pack, 'L': z.left, 'R': z.right}[dst]()
            assert_same_seq(out, orc.to_kind(osq[k], dst, fill, srt), f'{k}->{dst}')
        for s in (-20, -1, 3, 9, 1000003):
            assert_same_seq(z.roll(s), orc.roll(osq[k], s, srt), f'roll {k} {s}')
        assert_same_seq(z.rev(), orc.rev(osq[k], srt), f'rev {k}')
        assert np.array_equal(to_np(z.last()), orc.last(osq[k])), f'last {k}'
        h = z.head(1)
        assert_same_seq(h._replace(data=h.data.contiguous()), orc.head(osq[k], 1), f'head {k}')


@settings(deadline=None, max_examples=int(os.environ.get('RUA_HYP_EXAMPLES', 40)))
@given(lens=st.lists(st.integers(1, 40), min_size=1, max_size=40), h=st.integers(1, 70),
       dtype=st.sampled_from([torch.float32, torch.bfloat16, torch.float16, torch.float64]),
       name=st.sampled_from(['sum', 'mean', 'max', 'min', 'prod', 'logsumexp']), kind=st.sampled_from('CLPR'))
def test_reduce_any_width_dtype_layout(lens, h, dtype, name, kind):
    data, lt = _mk(lens, (h,), dtype)
    data = (data * 0.5).to(dtype)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    ref = getattr(orc, f'segment_{name}')(f, lt.numpy())                 # reference semantics on the upcast input
    z = {'C': lambda c: c, 'L': lambda c: c.left(), 'P': lambda c: c.pack(), 'R': lambda c: c.right()}[kind](
        ta.C(data.to(DEV), lt.to(DEV)))
    out = getattr(ta, f'reduce_{name}')(z)
    assert out.dtype == dtype and out.shape == (len(lens), h)
    # fp32 accumulate then ONE rounding to the output dtype: 1e-5 on the accumulation + half an ulp of the dtype
    ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.astype(np.float64), rtol=1e-5 + ulp, atol=1e-5 + ulp)


@pytest.mark.parametrize('h,dtype', [(70001, torch.float32), (65536, torch.bfloat16), (4100, torch.float64)])
def test_reduce_very_wide_rows(h, dtype):
    """Rows of hundreds of KiB (thousands of column chunks per sequence; odd widths take the scalar path)."""
    lens = [5, 1, 9, 2, 70]
    data, lt = _mk(lens, (h,), dtype, seed=h)
    data = (data * 0.5).to(dtype)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    c = ta.C(data.to(DEV), lt.to(DEV))
    for name in ('sum', 'mean', 'max', 'min', 'prod', 'logsumexp'):
        ref = getattr(orc, f'segment_{name}')(f, lt.numpy()).astype(np.float64)
        for z in (c, c.pack(), c.left()):
            out = getattr(ta, f'reduce_{name}')(z)
            np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=2e-5 + ulp, atol=2e-5 + ulp, err_msg=name)


@pytest.mark.parametrize('h,dtype', [(16, torch.bfloat16), (24, torch.float32), (3, torch.float32)])
def test_reduce_packed_many_narrow_sequences(h, dtype):
    """Enough narrow-row sequences that reduce(P) and its backward take the adjacent-rank kernels
    (B / ranks-per-wave >= 4096 waves): forward vs the oracle, backward vs the same op over the C layout."""
    B = 140_000 if h != 24 else 40_000
    g = torch.Generator().manual_seed(h)
    lt = torch.randint(1, 6, (B,), generator=g)
    data = (torch.randn(int(lt.sum()), h, generator=g) * 0.5).to(dtype)
    f = data.float().numpy()
    ulp = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    c = ta.with_host_sizes(data.to(DEV), lt)
    p = c.pack()
    for name in ('sum', 'mean', 'max', 'min', 'prod', 'logsumexp'):
        ref = getattr(orc, f'segment_{name}')(f, lt.numpy())
        xp = p.data.detach().clone().requires_grad_(True)
        out = getattr(ta, f'reduce_{name}')(p._replace(data=xp))
        np.testing.assert_allclose(out.detach().double().cpu().numpy(), ref.astype(np.float64), rtol=1e-5 + ulp,
                                   atol=1e-5 + ulp, err_msg=name)
        cot = torch.randn(out.shape, generator=g).to(dtype).to(DEV)
        out.backward(cot)
        xc = c.data.detach().clone().requires_grad_(True)
        getattr(ta, f'segment_{name}')(xc, c.token_sizes).backward(cot)
        # the gradient of the packed rows, brought back to C order, is the gradient over the C layout
        back = p._replace(data=xp.grad).cat().data
        torch.testing.assert_close(back.float(), xc.grad.float(), rtol=1e-5 + 4 * ulp, atol=1e-6 + 4 * ulp, msg=name)


@pytest.mark.parametrize('h,dtype', [(16, torch.bfloat16), (8, torch.float32), (4, torch.float32)])
def test_segment_reduce_many_narrow_segments_with_empties(h, dtype):
    """140 000 segments of 0..5 rows of at most 32 B in a CattedSequence , empty
    segments included: every op vs the oracle (the reference's global-extreme `initial` for the empty ones)."""
    g = torch.Generator().manual_seed(h + 100)
    lt = torch.randint(0, 6, (140_000,), generator=g)
    data = (torch.randn(int(lt.sum()), h, generator=g) * 0.5).to(dtype)
    f = data.float().numpy()
    ulp = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    for name in ('sum', 'mean', 'max', 'min', 'prod', 'logsumexp'):
        ref = getattr(orc, f'segment_{name}')(f, lt.numpy())
        out = getattr(ta, f'segment_{name}')(data.to(DEV), lt.to(DEV))
        np.testing.assert_allclose(out.double().cpu().numpy(), ref.astype(np.float64), rtol=1e-5 + ulp,
                                   atol=1e-5 + ulp, err_msg=name)


def test_zero_length_segments_and_initial():
    """Empty segments take the reference's `initial`: 0 / 1 / the GLOBAL min (max) resp. max (min)."""
    lens = [0, 3, 0, 0, 2, 5, 0]
    data, lt = _mk(lens, (6,), torch.float32, seed=3)
    for name in ('sum', 'mean', 'prod', 'max', 'min', 'logsumexp'):
        ref = getattr(orc, f'segment_{name}')(data.numpy(), lt.numpy())
        out = getattr(ta, f'segment_{name}')(data.to(DEV), lt.to(DEV)).cpu().numpy()
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6, err_msg=name)
    # zero-length sequences inside a CattedSequence survive pad / unpad (the reference's P cannot hold them)
    c = ta.C(data.to(DEV), lt.to(DEV))
    oc = orc.C(data.numpy(), lt.numpy())
    assert_same_seq(c.left(-2.0), orc.to_left(oc, -2.0), 'left with empties')
    assert_same_seq(c.right(-2.0), orc.to_right(oc, -2.0), 'right with empties')
    assert torch.equal(c.left().cat().data, c.data)
    assert torch.equal(c.roll(2).roll(-2).data, c.data)


def test_nan_semantics_match_reference():
    data, lt = _mk([2, 3, 1], (4,), torch.float32, seed=1)
    data[3, 2] = float('nan')
    for name in ('max', 'min', 'sum'):
        ref = getattr(orc, f'segment_{name}')(data.numpy(), lt.numpy())
        out = getattr(ta, f'segment_{name}')(data.to(DEV), lt.to(DEV)).cpu().numpy()
        assert np.array_equal(np.isnan(out), np.isnan(ref)), name     # incl. the reference's initial=NaN poisoning
        np.testing.assert_allclose(out[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('lens', [[2, 3, 1, 70, 4], [2, 0, 3, 1, 200, 0, 9]])
def test_infinities_nan_and_empty_segments_like_the_reference(lens, dtype):
    """max / min / logsumexp with -inf, +inf and NaN elements, all--inf sequences (NaN in the reference's logsumexp),
    empty segments (the reference's global-extreme `initial`) and NaN poisoning: same NaN pattern, same values."""
    import warnings
    g = torch.Generator().manual_seed(sum(lens))
    lt = torch.tensor(lens)
    base = torch.randn(int(lt.sum()), 16, generator=g).to(dtype)
    starts = torch.cumsum(lt, 0) - lt
    cases = {'clean': base.clone()}
    x = base.clone(); x[int(starts[0]):int(starts[0]) + lens[0], 3] = float('-inf'); cases['one column of a sequence all -inf'] = x
    x = base.clone(); x[int(starts[2]) + 1, 5] = float('inf'); cases['+inf element'] = x
    x = base.clone(); x[int(starts[3]) + 2, :] = float('-inf'); cases['a -inf row'] = x
    x = base.clone(); x[int(starts[4]), 7] = float('nan'); cases['NaN element'] = x
    for what, data in cases.items():
        f = data.float().numpy()
        for name in ('max', 'min', 'logsumexp'):
            with warnings.catch_warnings(), np.errstate(all='ignore'):
                warnings.simplefilter('ignore')
                ref = getattr(orc, f'segment_{name}')(f, lt.numpy())
            out = getattr(ta, f'segment_{name}')(data.to(DEV), lt.to(DEV)).float().cpu().numpy()
            assert np.array_equal(np.isnan(out), np.isnan(ref)), (what, name)
            ok = ~np.isnan(ref)
            ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 0.0
            np.testing.assert_allclose(out[ok], ref[ok], rtol=1e-5 + ulp, atol=1e-6, err_msg=f'{what} {name}')
            if 0 not in lens:     # the same through a PackedSequence (zero-length sequences do not pack)
                pk = getattr(ta, f'reduce_{name}')(ta.C(data.to(DEV), lt.to(DEV)).pack()).float().cpu().numpy()
                assert np.array_equal(np.isnan(pk), np.isnan(ref)), (what, name, 'P')
                np.testing.assert_allclose(pk[ok], ref[ok], rtol=1e-5 + ulp, atol=1e-6, err_msg=f'{what} {name} P')


def test_max_min_backward_counts_come_from_the_forward():
    """The differentiable forward of max/min also counts the elements equal to the extreme (so the backward is one
    walk): ties inside one chunk, across chunks and row groups, every layout, split sequences — against autograd through torch.segment_reduce on the CPU, the reference's own backend call
    (reduce.py:34-41), with cotangents of BOTH signs: its backward lets ties share a positive gradient and hands each
    of them a non-positive one whole."""
    g = torch.Generator().manual_seed(21)
    lens = [1, 300, 7, 64, 65, 5000, 2, 129]
    lt = torch.tensor(lens)
    for h, dtype in ((8, torch.float32), (24, torch.float32), (512, torch.float32), (130, torch.float64)):
        x = torch.randint(0, 3, (sum(lens), h), generator=g).to(dtype)
        cot = torch.randn(len(lens), h, generator=g).to(dtype)
        cot[:, 0] = 0.0                                         # and a zero gradient
        for name in ('max', 'min'):
            r = x.clone().requires_grad_(True)
            ref = torch.segment_reduce(r, name, lengths=lt, unsafe=True)
            ref.backward(cot)
            for kind in 'CLPR':
                xs = x.clone().to(DEV).requires_grad_(True)
                c = ta.with_host_sizes(xs, lt)
                z = {'C': lambda: c, 'L': c.left, 'P': c.pack, 'R': c.right}[kind]()
                out = getattr(ta, f'reduce_{name}')(z)
                out.backward(cot.to(DEV))
                torch.testing.assert_close(out.detach().cpu(), ref.detach())
                torch.testing.assert_close(xs.grad.cpu(), r.grad, rtol=1e-6, atol=1e-7, msg=f'{name} {kind} h={h}')
    # many short sequences with narrow rows (a PackedSequence then takes the adjacent-ranks kernel), signed cotangents
    lt3 = torch.randint(1, 12, (3000,), generator=g)
    x3 = torch.randint(0, 3, (int(lt3.sum()), 4), generator=g).float()
    cot3 = torch.randn(3000, 4, generator=g)
    for name in ('max', 'min'):
        r = x3.clone().requires_grad_(True)
        torch.segment_reduce(r, name, lengths=lt3, unsafe=True).backward(cot3)
        for kind in 'CLPR':
            xs = x3.clone().to(DEV).requires_grad_(True)
            c = ta.with_host_sizes(xs, lt3)
            z = {'C': lambda: c, 'L': c.left, 'P': c.pack, 'R': c.right}[kind]()
            getattr(ta, f'reduce_{name}')(z).backward(cot3.to(DEV))
            torch.testing.assert_close(xs.grad.cpu(), r.grad, rtol=1e-6, atol=1e-7, msg=f'narrow {name} {kind}')
    # a NaN result: its NaN elements are the hits and share the gradient (column 0; every sequence holds one there, so
    # the reference's NaN-poisoned `initial` changes nothing in that column)
    x = torch.tensor([[float('nan'), 1.0], [1.0, 2.0], [float('nan'), 3.0], [float('nan'), 4.0], [5.0, 0.0]])
    lt2 = torch.tensor([3, 2])
    r = x.clone().requires_grad_(True)
    torch.segment_reduce(r, 'max', lengths=lt2, unsafe=True).backward(torch.ones(2, 2))
    xs = x.clone().to(DEV).requires_grad_(True)
    ta.segment_max(xs, lt2.to(DEV)).backward(torch.ones(2, 2, device=DEV))
    assert xs.grad[:, 0].tolist() == r.grad[:, 0].tolist() == [0.5, 0.0, 0.5, 1.0, 0.0]


def test_non_contiguous_and_views():
    data, lt = _mk([4, 2, 5, 1], (8,), torch.float32)
    wide = torch.randn(12, 16, device=DEV)
    wide[:, ::2] = data.to(DEV)
    c = ta.C(wide[:, ::2], lt.to(DEV))                                  # strided payload
    oc = orc.C(data.numpy(), lt.numpy())
    assert_same_seq(c.pack(), orc.to_pack(oc, host_sort(lt)), 'pack of a strided view')
    l = c.left()
    assert torch.equal(l.head(2).cat().data, c.head(2).data)            # L.head is a view; cat() of a view
    assert torch.equal(l.trunc((1, 0)).cat().data, c.trunc((1, 0)).data) if min(lt) > 1 else True
    p = c.pack()
    assert torch.equal(p.head(1).cat().data, c.head(1).data)            # P.head is a view sharing indices


def test_single_sequences_and_scalars():
    for lens, hidden in (([1], ()), ([1], (1,)), ([9], (2,)), ([1, 1, 1], (3,))):
        data, lt = _mk(lens, hidden, torch.float32)
        c = ta.C(data.to(DEV), lt.to(DEV))
        oc = orc.C(data.numpy(), lt.numpy())
        srt = host_sort(lens)
        assert_same_seq(c.pack(), orc.to_pack(oc, srt))
        assert_same_seq(c.pack().left(), orc.to_left(oc))
        assert_same_seq(c.right().pack().cat(), oc)
        assert np.array_equal(to_np(c.last()), orc.last(oc))
        assert np.allclose(to_np(ta.reduce_sum(c.pack())), orc.segment_sum(data.numpy().reshape(len(data), -1), lt.numpy()).reshape(
            (len(lens),) + tuple(hidden)), atol=1e-6)


def test_index_primitives():
    sizes = torch.tensor([3, 0, 2, 5, 1], device=DEV)
    assert ta.get_offsets(sizes).tolist() == [0, 3, 3, 5, 10]
    major, minor = ta.major_sizes_to_ptr(sizes)
    assert major.tolist() == [0, 1, 2, 0, 1, 0, 1, 2, 3, 4, 0] and minor.tolist() == [0, 0, 0, 2, 2, 3, 3, 3, 3, 3, 4]
    perm = torch.randperm(1000, device=DEV)
    inv = ta.invert_permutation(perm)
    assert torch.equal(inv[perm], torch.arange(1000, device=DEV))
    big = torch.randint(0, 50, (300007,), device=DEV)
    assert torch.equal(ta.get_offsets(big), torch.cumsum(big, 0) - big)


def test_scatter_large_fan_in_is_deterministic():
    """Buckets longer than a wave (rank-by-counting path) and run-to-run bitwise reproducibility."""
    g = torch.Generator().manual_seed(0)
    S, M, H = 7, 5000, 16
    index = torch.randint(0, S, (M,), generator=g).to(DEV)
    src = torch.randn(M, H, generator=g).to(DEV)
    ten = torch.randn(S, H, generator=g).to(DEV)
    a = ta.scatter_sum(ten, index, src, include_self=True)
    b = ta.scatter_sum(ten, index, src, include_self=True)
    assert torch.equal(a, b)
    ref = orc.scatter_sum(ten.cpu().numpy(), index.cpu().numpy(), src.cpu().numpy(), include_self=True)
    np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    m = ta.scatter_max(ten, index, src)
    assert np.array_equal(m.cpu().numpy(), orc.scatter_max(ten.cpu().numpy(), index.cpu().numpy(), src.cpu().numpy()))


def test_backward_ties_and_long_sequences():
    """max/min share the gradient equally among ties (torch.segment_reduce backward; SURVEY.md §4: [1,5,5] ->
    [0,.5,.5]); sequences longer than the kernel's 64-row table block; every layout."""
    x = torch.tensor([[1.0], [5.0], [5.0], [2.0], [2.0], [2.0]], device=DEV, requires_grad=True)
    lens = torch.tensor([3, 3], device=DEV)
    ta.segment_max(x, lens).sum().backward()
    third = float(torch.tensor(1.0) / 3)
    assert x.grad.view(-1).tolist() == [0.0, 0.5, 0.5, third, third, third]
    x.grad = None
    ta.segment_min(x, lens).sum().backward()
    assert x.grad.view(-1).tolist() == [1.0, 0.0, 0.0, third, third, third]

    g = torch.Generator().manual_seed(0)
    lens = torch.tensor([200, 1, 131, 64, 65])
    data = torch.randn(int(lens.sum()), 24, generator=g)
    for name, fn in (('sum', lambda t: t.sum(0)), ('mean', lambda t: t.mean(0)), ('max', lambda t: t.max(0).values),
                     ('logsumexp', lambda t: t.logsumexp(0)), ('prod', lambda t: (t * 0.9).prod(0))):
        for kind in 'CLPR':
            xs = data.clone().to(DEV).requires_grad_(True)
            scaled = xs * 0.9 if name == 'prod' else xs
            c = ta.C(scaled, lens.to(DEV))
            z = {'C': c, 'L': c.left, 'P': c.pack, 'R': c.right}[kind]
            z = z if kind == 'C' else z()
            out = getattr(ta, f'reduce_{name}')(z)
            cot = torch.randn(out.shape, generator=g).to(DEV)
            out.backward(cot)
            ref_in = data.clone().to(DEV).requires_grad_(True)
            ref = torch.stack([fn(s) for s in torch.split(ref_in, lens.tolist())])
            ref.backward(cot)
            torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(xs.grad, ref_in.grad, rtol=1e-4, atol=1e-5, msg=f'{name} {kind}')


@pytest.mark.parametrize('hidden,dtype', [(8, torch.float32), (512, torch.bfloat16), (70, torch.float32), (130, torch.float64)])
def test_long_sequences_are_split(hidden, dtype):
    """Sequences far beyond the reducer's 4 096-row part size (split + tail + combine kernels), next to
    short ones and exact multiples; host-known and device-only lengths; every op; C / P / L inputs; fused."""
    from torchrua_amd import _meta as M
    lens = [20000, 3, 9000, 1, 4097, 4096, 12289, 8192, 257, 256, 511]
    g = torch.Generator().manual_seed(4)
    data = (torch.randn(sum(lens), hidden, generator=g) * 0.1).to(dtype)
    lt = torch.tensor(lens)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    known = ta.with_host_sizes(data.to(DEV), lt)          # host mirror: the split is chosen from max(len)
    blind = ta.C(data.to(DEV), lt.to(DEV))                # device-only lengths: armed by the heuristic
    rb = hidden * data.element_size()
    part = M.reduce_split_rows(M.lay_cat(known.token_sizes, len(lens), sum(lens)), rb)
    assert 0 < part < 20000 and part == M.reduce_split_rows(M.lay_cat(blind.token_sizes, len(lens), sum(lens)), rb)
    p = known.pack()
    for name in ('sum', 'mean', 'max', 'min', 'logsumexp', 'prod'):
        src = f if name != 'prod' else np.where(np.abs(f) > 0, 1.0 + f * 1e-3, 1.0).astype(f.dtype)
        dd = data if name != 'prod' else torch.from_numpy(src).to(dtype)
        up = dd.double().numpy() if dtype == torch.float64 else dd.float().numpy()   # what the kernel really sees
        ref = getattr(orc, f'segment_{name}')(up, lt.numpy())
        kc = ta.with_host_sizes(dd.to(DEV), lt)
        outs = {'segment(known)': getattr(ta, f'segment_{name}')(kc.data, kc.token_sizes),
                'segment(blind)': getattr(ta, f'segment_{name}')(dd.to(DEV), blind.token_sizes),
                'reduce(P)': getattr(ta, f'reduce_{name}')(kc.pack())}
        if hidden % (16 // data.element_size()) == 0:
            pf, of = ta.pack_reduce(kc, name, fused=True)
            outs['fused'] = of
            assert torch.equal(pf.data, kc.pack().data)
        if hidden <= 70:
            outs['reduce(L)'] = getattr(ta, f'reduce_{name}')(kc.left())
        for what, got in outs.items():
            scale = 20000 * 0.5 if name == 'sum' else 1.0
            # a 20 000-factor fp32 product carries ~n*2^-24 of rounding on BOTH sides (order differs): 1e-3
            rtol = 1e-3 if name == 'prod' else 2e-5
            np.testing.assert_allclose(got.double().cpu().numpy(), ref.astype(np.float64), rtol=rtol + ulp,
                                       atol=1e-5 * scale + ulp, err_msg=f'{what} {name}')
    # gradients through split sequences (the backward publishes parts too)
    xf = torch.randn(sum(lens), hidden, generator=g) * 0.1     # fresh fp32 values: no ties for max
    for name, fn in (('sum', lambda t: t.sum(0)), ('mean', lambda t: t.mean(0)), ('logsumexp', lambda t: t.logsumexp(0)),
                     ('max', lambda t: t.max(0).values)):
        x = xf.clone().to(DEV).requires_grad_(True)
        out = getattr(ta, f'segment_{name}')(x, known.token_sizes)
        cot = torch.randn(out.shape, generator=g).to(DEV)
        out.backward(cot)
        r = xf.clone().to(DEV).requires_grad_(True)
        torch.stack([fn(s_) for s_ in torch.split(r, lens)]).backward(cot)
        torch.testing.assert_close(x.grad, r.grad, rtol=1e-4, atol=1e-6, msg=f'backward {name}')
    # max/min with ties spread over several parts of a split sequence (phased backward: count, then apply)
    xt = torch.randint(0, 3, (sum(lens), hidden), generator=g).float()
    for name in ('max', 'min'):
        x = xt.clone().to(DEV).requires_grad_(True)
        out = getattr(ta, f'segment_{name}')(x, known.token_sizes)
        cot = torch.randn(out.shape, generator=g)
        out.backward(cot.to(DEV))
        r = xt.clone().requires_grad_(True)                      # the reference's backend call, on the CPU
        torch.segment_reduce(r, name, lengths=torch.tensor(lens), unsafe=True).backward(cot)
        torch.testing.assert_close(x.grad.cpu(), r.grad, rtol=1e-5, atol=1e-7, msg=f'tied backward {name}')
    # prod with a single zero factor deep inside the longest sequence: its gradient is the product of the others
    xp = 1.0 + torch.randn(sum(lens), hidden, generator=g) * 1e-3
    xp[12345] = 0.0
    x = xp.clone().to(DEV).requires_grad_(True)
    ta.segment_prod(x, known.token_sizes).sum().backward()
    r = xp.clone().to(DEV).requires_grad_(True)
    torch.stack([s_.prod(0) for s_ in torch.split(r, lens)]).sum().backward()
    torch.testing.assert_close(x.grad, r.grad, rtol=1e-3, atol=1e-6, msg='prod backward with a zero factor')
    assert float(x.grad[12345].abs().min()) > 0.5 and float(x.grad[:20000].abs().sum(0).min()) == float(x.grad[12345].abs().min())
    # scatter_max/min with include_self over huge buckets: both gradients vs torch.index_reduce
    idx_t = torch.repeat_interleave(torch.arange(len(lens)), lt)[torch.randperm(sum(lens), generator=g)].to(DEV)
    for name, red in (('max', 'amax'), ('min', 'amin')):
        for inc in (True, False):
            a = torch.randint(0, 3, (len(lens), hidden), generator=g).float().to(DEV).requires_grad_(True)
            s_ = xt.clone().to(DEV).requires_grad_(True)
            out = getattr(ta, f'scatter_{name}')(a, idx_t, s_, include_self=inc)
            cot = torch.randn(out.shape, generator=g).to(DEV)
            out.backward(cot)
            a2, s2 = a.detach().clone().requires_grad_(True), s_.detach().clone().requires_grad_(True)
            ref = a2.index_reduce(0, idx_t, s2, red, include_self=inc)
            ref.backward(cot)
            assert torch.equal(out, ref)
            torch.testing.assert_close(s_.grad, s2.grad, rtol=1e-5, atol=1e-7, msg=f'scatter_{name} src inc={inc}')
            torch.testing.assert_close(a.grad, a2.grad, rtol=1e-5, atol=1e-7, msg=f'scatter_{name} self inc={inc}')
    # scatter with one huge destination: forward and backward both go through the split path
    idx = torch.repeat_interleave(torch.arange(len(lens)), lt)
    perm = torch.randperm(idx.numel(), generator=g)
    src = xf[perm].clone().to(DEV).requires_grad_(True)
    ten = torch.zeros(len(lens), hidden, device=DEV)
    ta.scatter_sum(ten, idx[perm].to(DEV), src).sum().backward()
    assert torch.equal(src.grad, torch.ones_like(src))


@pytest.mark.parametrize('B,hi', [(1, 1), (300, 40), (5000, 2048), (32768, 9), (32769, 9), (70000, 3), (50, 2049), (65536, 600),
                                  (131072, 3), (131073, 3), (2048, 5), (2049, 5), (4097, 2)])
def test_pack_prepare_equals_the_separate_steps(B, hi):
    """rua_pack_prepare (one launch up to T = 2 048 / B = 131 072 — a block per tile of the lengths —, three steps
    beyond) vs torch on the host."""
    g = torch.Generator().manual_seed(B + hi)
    lens = torch.randint(1, hi + 1, (B,), generator=g)
    lens[int(torch.randint(0, B, (1,), generator=g))] = hi
    c = ta.with_host_sizes(torch.zeros(int(lens.sum()), 1, device=DEV), lens)
    p = c.pack()
    srt = torch.sort(lens, descending=True)[1]
    assert torch.equal(p.sorted_indices.cpu(), srt)
    assert torch.equal(p.unsorted_indices.cpu(), torch.argsort(srt))
    bsz = (lens[None, :] > torch.arange(hi)[:, None]).sum(1)
    assert torch.equal(p.batch_sizes, bsz)
    from torchrua_amd import _meta as M
    assert torch.equal(M.pack_boff(p).cpu(), torch.cumsum(bsz, 0) - bsz)
    assert torch.equal(M.dev_off(c.token_sizes).cpu(), torch.cumsum(lens, 0) - lens)


def test_bucketing_is_a_stable_sort():
    """rua_index_buckets == a stable sort of the rows by destination (checked against torch.sort(stable=True)),
    for few / many destinations, 1-3 radix passes, out-of-range indices ignored."""
    from torchrua_amd.reduce import _buckets
    g = torch.Generator().manual_seed(1)
    for S, M in ((1, 10), (7, 5000), (255, 70000), (257, 70000), (65536, 300000), (70000, 123457), (3, 2049)):
        index = torch.randint(0, S, (M,), generator=g).to(DEV)
        counts, perm = _buckets(index, S)
        order = torch.sort(index, stable=True)[1]
        assert torch.equal(perm, order), (S, M)
        assert torch.equal(counts, torch.bincount(index, minlength=S)), (S, M)
    index = torch.tensor([2, -1, 0, 9, 2, 0, 5], device=DEV)          # -1, 9, 5 are out of range for S = 3
    counts, perm = _buckets(index, 3)
    assert counts.tolist() == [2, 0, 2] and perm[:4].tolist() == [2, 5, 0, 4]


def test_scatter_huge_fan_in_is_fast_and_right():
    """One destination receiving 300 000 rows: deterministic order, right values, twice bit-identical."""
    import time
    g = torch.Generator().manual_seed(0)
    M, H = 300_000, 8
    index = torch.zeros(M, dtype=torch.long)
    index[::1000] = 1
    src = torch.randn(M, H, generator=g)
    ten = torch.zeros(2, H)
    t0 = time.perf_counter()
    out = ta.scatter_sum(ten.to(DEV), index.to(DEV), src.to(DEV))
    mx = ta.scatter_max(ten.to(DEV), index.to(DEV), src.to(DEV))
    assert torch.equal(out, ta.scatter_sum(ten.to(DEV), index.to(DEV), src.to(DEV)))   # bitwise reproducible
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0
    ref = torch.zeros(2, H, dtype=torch.float64).index_add_(0, index, src.double())
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), rtol=1e-3, atol=2e-2)
    assert torch.equal(mx.cpu()[0], src[index == 0].max(0).values) and torch.equal(mx.cpu()[1], src[index == 1].max(0).values)


def test_empty_batch_and_empty_rows():
    """Degenerate shapes go through without touching the GPU with bad geometry."""
    e = ta.C(torch.empty(0, 4, device=DEV), torch.empty(0, dtype=torch.long, device=DEV))
    assert e.cat() is e
    assert ta.segment_sum(e.data, e.token_sizes).shape == (0, 4)
    assert e.roll(1).data.shape == (0, 4) and e.head(1).data.shape == (0, 4)
    p = e.pack()
    assert p.data.shape == (0, 4) and p.batch_sizes.numel() == 0
    assert e.left().data.shape[0] == 0
    assert ta.get_offsets(e.token_sizes).numel() == 0
    z = ta.C(torch.randn(5, 0, device=DEV), torch.tensor([2, 3], device=DEV))      # rows of zero width
    assert z.pack().data.shape == (5, 0) and z.left().data.shape == (2, 3, 0)
    assert ta.reduce_sum(z).shape == (2, 0)
    only_empty = ta.C(torch.empty(0, 3, device=DEV), torch.zeros(4, dtype=torch.long, device=DEV))
    assert torch.equal(ta.segment_sum(only_empty.data, only_empty.token_sizes), torch.zeros(4, 3, device=DEV))
    assert torch.equal(ta.segment_prod(only_empty.data, only_empty.token_sizes), torch.ones(4, 3, device=DEV))


def test_inconsistent_metadata_cannot_fault_the_gpu():
    """Lengths that sum past the payload, or a PackedSequence whose indices are garbage, read as padding /
    nothing instead of going out of bounds (a GPU fault here can reset the whole node)."""
    data = torch.randn(10, 16, device=DEV)
    bad = ta.C(data, torch.tensor([6, 9, 4], device=DEV))                # sums to 19 > 10 rows
    out = ta.segment_sum(data, bad.token_sizes)
    assert out.shape == (3, 16) and torch.isfinite(out).all()
    torch.testing.assert_close(out[0], data[:6].sum(0))
    left = bad.left(-1.0)
    assert left.data.shape == (3, 9, 16) and torch.equal(left.data[0, :6], data[:6])
    good = ta.C(data, torch.tensor([3, 5, 2], device=DEV)).pack()
    broken = good._replace(sorted_indices=torch.tensor([7, -3, 99], device=DEV))
    assert broken.cat().data.shape == (10, 16)                            # no fault; contents unspecified
    assert ta.reduce_sum(broken).shape == (3, 16)
    key = (torch.tensor([0, 5, -1], device=DEV), torch.tensor([1, 0, 2], device=DEV))
    got = ta.C(data, torch.tensor([3, 5, 2], device=DEV))[key]            # batch_ptr out of range -> zero rows
    assert torch.equal(got[0], data[1]) and bool((got[1] == 0).all()) and bool((got[2] == 0).all())
    torch.cuda.synchronize()


def test_prod_backward_with_zeros_matches_torch():
    """d prod / d x_i = prod of the OTHER factors, also when x_i == 0 (torch.segment_reduce special-cases zeros)."""
    x = torch.tensor([[2.0, 0.0, 3.0, 0.0], [0.0, 5.0, 4.0, 0.0], [3.0, 2.0, 0.0, 7.0],        # seq 0 (3 rows)
                      [1.5, 2.5, 3.5, 4.5], [0.0, 0.0, 1.0, 2.0]], device=DEV)                  # seq 1 (2 rows)
    lens = torch.tensor([3, 2], device=DEV)
    for kind in 'CLPR':
        a = x.clone().requires_grad_(True)
        c = ta.C(a, lens)
        z = {'C': lambda: c, 'L': c.left, 'P': c.pack, 'R': c.right}[kind]()
        out = ta.reduce_prod(z)
        cot = torch.tensor([[1.0, 2.0, 3.0, 4.0], [0.5, 1.5, 2.5, 3.5]], device=DEV)
        out.backward(cot)
        b = x.clone().requires_grad_(True)
        ref = torch.stack([b[:3].prod(0), b[3:].prod(0)])
        ref.backward(cot)
        torch.testing.assert_close(out, ref)
        torch.testing.assert_close(a.grad, b.grad, msg=kind)
    # and through the reference's own op on the CPU
    r = x.cpu().clone().requires_grad_(True)
    torch.segment_reduce(r, 'prod', lengths=lens.cpu(), unsafe=True, initial=1).backward(cot.cpu())
    torch.testing.assert_close(a.grad.cpu(), r.grad)
