"""Property tests on the GPU in the style of the reference's own suite (hypothesis over
random ragged batches, every layout, values AND gradients), with expectations built from stock torch
ops that do not touch torchrua_amd: pack_sequence / pad_sequence / per-sequence slicing and reductions."""
import os

import pytest
import torch
from hypothesis import given, settings, strategies as st
from torch.nn.utils.rnn import pack_sequence, pad_sequence

import torchrua_amd as ta
from gpu_util import DEV

pytestmark = pytest.mark.gpu

LAYOUTS = [ta.C, ta.L, ta.P, ta.R]
lens_st = st.lists(st.integers(1, 23), min_size=1, max_size=19)
dim_st = st.integers(1, 33)
SET = dict(deadline=None, max_examples=int(os.environ.get('RUA_HYP_EXAMPLES', 30)))   # soak: RUA_HYP_EXAMPLES=500


def make(lens, dim, dtype=torch.float32):
    return [torch.randn((n, dim), device=DEV, dtype=dtype, requires_grad=True) for n in lens]


def right_pad(seqs, value=0.0):
    t = max(s.size(0) for s in seqs)
    return torch.stack([torch.nn.functional.pad(s, [0, 0, t - s.size(0), 0], value=value) for s in seqs])


def grads(out, inputs, cot):
    return torch.autograd.grad(out, inputs, cot, allow_unused=True, retain_graph=True)


def check_grads(actual, expected, inputs):
    cot = torch.randn_like(expected)
    ga, ge = grads(actual, inputs, cot), grads(expected, inputs, cot)
    for a, e, x in zip(ga, ge, inputs):
        a = torch.zeros_like(x) if a is None else a
        e = torch.zeros_like(x) if e is None else e
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)


@settings(**SET)
@given(lens=lens_st, dim=dim_st, src=st.sampled_from(LAYOUTS))
def test_to_cat(lens, dim, src):
    xs = make(lens, dim)
    out = src.new(xs).cat()
    exp = torch.cat(xs)
    assert torch.equal(out.data, exp) and out.token_sizes.tolist() == lens
    check_grads(out.data, exp, xs)


@settings(**SET)
@given(lens=lens_st, dim=dim_st, src=st.sampled_from(LAYOUTS), fill=st.sampled_from([0.0, -3.5]))
def test_to_left_right(lens, dim, src, fill):
    xs = make(lens, dim)
    z = src.new(xs)
    left, right = z.left(fill), z.right(fill)
    # L.left / R.right are the identity (reference core/cast.py:33,69): they keep the constructor's padding (0)
    exp_l = pad_sequence(xs, batch_first=True, padding_value=0.0 if src is ta.L else fill)
    exp_r = right_pad(xs, 0.0 if src is ta.R else fill)
    assert torch.equal(left.data, exp_l) and torch.equal(right.data, exp_r)
    assert left.token_sizes.tolist() == lens == right.token_sizes.tolist()
    check_grads(left.data, exp_l, xs)
    check_grads(right.data, exp_r, xs)


@settings(**SET)
@given(lens=lens_st, dim=dim_st, src=st.sampled_from(LAYOUTS))
def test_to_pack(lens, dim, src):
    xs = make(lens, dim)
    out = src.new(xs).pack()
    exp = pack_sequence(xs, enforce_sorted=False)        # the same host sort as the reference
    assert torch.equal(out.data, exp.data)
    assert torch.equal(out.batch_sizes, exp.batch_sizes)
    assert torch.equal(out.sorted_indices, exp.sorted_indices)
    assert torch.equal(out.unsorted_indices, exp.unsorted_indices)
    check_grads(out.data, exp.data, xs)


@settings(**SET)
@given(data=st.data(), lens=lens_st, dim=dim_st, src=st.sampled_from(LAYOUTS))
def test_head_last_trunc(data, lens, dim, src):
    xs = make(lens, dim)
    n = data.draw(st.integers(1, min(lens)))
    out = src.new(xs).head(n).cat()
    exp = torch.cat([x[:n] for x in xs])
    assert torch.equal(out.data, exp) and out.token_sizes.tolist() == [n] * len(lens)
    check_grads(out.data, exp, xs)

    last = src.new(xs).last()
    exp = torch.stack([x[-1] for x in xs])
    assert torch.equal(last, exp)
    check_grads(last, exp, xs)

    a = data.draw(st.integers(0, min(lens) - 1))
    b = data.draw(st.integers(0, min(lens) - 1 - a))
    out = src.new(xs).trunc((a, b)).cat()
    exp = torch.cat([x[a:x.size(0) - b] for x in xs])
    assert torch.equal(out.data, exp) and out.token_sizes.tolist() == [m - a - b for m in lens]
    check_grads(out.data, exp, xs)


@settings(**SET)
@given(data=st.data(), lens=lens_st, dim=dim_st, src=st.sampled_from(LAYOUTS))
def test_roll_rev(data, lens, dim, src):
    xs = make(lens, dim)
    s = data.draw(st.integers(-max(lens) - 2, max(lens) + 2))
    out = src.new(xs).roll(s).cat()
    exp = torch.cat([x.roll(s, dims=[0]) for x in xs])
    assert torch.equal(out.data, exp) and out.token_sizes.tolist() == lens
    check_grads(out.data, exp, xs)
    out = src.new(xs).rev().cat()
    exp = torch.cat([x.flip(dims=[0]) for x in xs])
    assert torch.equal(out.data, exp)
    check_grads(out.data, exp, xs)


REDUCERS = {
    'sum': lambda x: x.sum(0), 'mean': lambda x: x.mean(0), 'max': lambda x: x.max(0).values,
    'min': lambda x: x.min(0).values, 'prod': lambda x: x.prod(0), 'logsumexp': lambda x: x.logsumexp(0),
}


@settings(**SET)
@given(lens=st.lists(st.integers(1, 9), min_size=1, max_size=15), dim=dim_st, name=st.sampled_from(sorted(REDUCERS)),
       src=st.sampled_from(LAYOUTS))
def test_reduce_any_layout(lens, dim, name, src):
    """segment_* over C and reduce_* straight over C/L/P/R (no conversion) vs per-sequence torch."""
    xs = make(lens, dim)
    exp = torch.stack([REDUCERS[name](x) for x in xs])
    c = ta.C.new(xs)
    seg = getattr(ta, f'segment_{name}')(c.data, c.token_sizes)
    torch.testing.assert_close(seg, exp, rtol=1e-5, atol=1e-5)
    red = getattr(ta, f'reduce_{name}')(src.new(xs))
    torch.testing.assert_close(red, exp, rtol=1e-5, atol=1e-5)
    check_grads(seg, exp, xs)
    check_grads(red, exp, xs)


@settings(**SET)
@given(lens=st.lists(st.integers(1, 7), min_size=1, max_size=15), dim=dim_st, name=st.sampled_from(sorted(REDUCERS)),
       include_self=st.booleans())
def test_scatter(lens, dim, name, include_self):
    xs = make(lens, dim)
    index = torch.cat([torch.full((n,), i, device=DEV) for i, n in enumerate(lens)])
    perm = torch.randperm(sum(lens), device=DEV)
    tensor = torch.randn((len(lens), dim), device=DEV, requires_grad=True)
    rows = [torch.cat([x, t[None]]) if include_self else x for x, t in zip(xs, tensor)]
    exp = torch.stack([REDUCERS[name](r) for r in rows])
    src = torch.cat(xs)
    out = getattr(ta, f'scatter_{name}')(tensor, index[perm], src[perm], include_self=include_self)
    torch.testing.assert_close(out, exp, rtol=1e-5, atol=1e-5)
    check_grads(out, exp, xs + ([tensor] if include_self else []))


@settings(**SET)
@given(lens=st.lists(st.integers(1, 12), min_size=1, max_size=9), dim=st.integers(1, 9),
       name=st.sampled_from(['sum', 'max', 'mean', 'logsumexp', 'min', 'prod']),
       seq=st.sampled_from(LAYOUTS), dur=st.sampled_from(LAYOUTS))
def test_seg(lens, dim, name, seq, dur):
    xs = make(lens, dim)
    durations = [torch.unique(torch.randint(n, (n,), device=DEV), return_counts=True)[1] for n in lens]
    out = seq.new(xs).seg(dur.new(durations), getattr(ta, f'segment_{name}')).cat()
    exp = []
    for x, d in zip(xs, durations):
        lo = 0
        for n in d.tolist():
            exp.append(REDUCERS[name](x[lo:lo + n]))
            lo += n
    exp = torch.stack(exp)
    assert out.token_sizes.tolist() == [d.numel() for d in durations]
    torch.testing.assert_close(out.data, exp, rtol=1e-5, atol=1e-5)
    check_grads(out.data, exp, xs)


@settings(**SET)
@given(lens=lens_st, src=st.sampled_from(LAYOUTS),
       spec=st.sampled_from([(False, True, torch.bool), (-1, 2, torch.long),
                             (torch.finfo(torch.float16).min, torch.finfo(torch.float16).max, torch.float16),
                             (torch.finfo(torch.float64).min, torch.finfo(torch.float64).max, torch.float64)]))
def test_mask(lens, src, spec):
    zero, one, dtype = spec
    xs = [torch.randn((n,), device=DEV) for n in lens]
    out = src.new(xs).mask(zero=zero, one=one, dtype=dtype)
    exp = pad_sequence([torch.full((n,), one, device=DEV, dtype=dtype) for n in lens], batch_first=True,
                       padding_value=zero)
    assert torch.equal(out, exp)


@settings(deadline=None, max_examples=10)
@given(groups=st.lists(st.lists(st.integers(1, 6), min_size=1, max_size=4), min_size=1, max_size=4), data=st.data())
def test_compose_and_split(groups, data):
    """compose == pack_sequence of per-container results (checked through an LSTM, like the reference's test)."""
    dim, hid = 3, 4
    rnn = torch.nn.LSTM(dim, hid, bidirectional=True).to(DEV)
    seqs = [[torch.randn((n, dim), device=DEV) for n in g] for g in groups]
    containers = [data.draw(st.sampled_from(LAYOUTS)).new(s) for s in seqs]
    _, (h, _) = rnn(ta.compose(containers))
    got = h.transpose(-3, -2).flatten(start_dim=-2)
    exp = []
    for s in seqs:
        _, (h, _) = rnn(pack_sequence(s, enforce_sorted=False))
        exp.append(h.transpose(-3, -2).flatten(start_dim=-2))
    exp = pack_sequence(exp, enforce_sorted=False).data
    torch.testing.assert_close(got, exp, rtol=1e-4, atol=1e-5)
    for c, s in zip(containers, seqs):
        for a, e in zip(c.split(), s):
            assert torch.equal(a, e)
        assert c.tolist() == [e.tolist() for e in s]


@settings(**SET)
@given(lens=lens_st, dim=st.sampled_from([4, 8, 16, 36, 3]), name=st.sampled_from(sorted(REDUCERS)),
       src=st.sampled_from([ta.C, ta.L, ta.R]), dtype=st.sampled_from([torch.float32, torch.bfloat16]))
def test_fused_pack_reduce_equals_two_calls(lens, dim, name, src, dtype):
    """pack_reduce (one pass) == pack() then reduce_*() bit for bit, PackedSequence metadata included."""
    xs = [torch.randn((n, dim), device=DEV).to(dtype) for n in lens]
    z = src.new(xs)
    p1 = z.pack()
    o1 = getattr(ta, f'reduce_{name}')(p1)
    p2, o2 = ta.pack_reduce(z, name, fused=True)
    assert torch.equal(p1.data, p2.data) and torch.equal(p1.batch_sizes, p2.batch_sizes)
    assert torch.equal(p1.sorted_indices, p2.sorted_indices) and torch.equal(p1.unsorted_indices, p2.unsorted_indices)
    assert torch.equal(o1, o2) or (torch.isnan(o1) == torch.isnan(o2)).all()
