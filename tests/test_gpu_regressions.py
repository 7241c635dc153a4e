"""Regressions for defects found by review: zero-length sequences through pack() at row
widths on both sides of the 128-byte kernel switch, scatter_* argument validation (what torch.index_add /
index_reduce reject), the hot path under torch.inference_mode(), scatter_prod's gradient where `tensor` is 0,
and the privacy of the host length mirror."""
import os

import numpy as np
import pytest
import torch

import torchrua_amd as ta
from gpu_util import DEV, assert_same_seq, host_sort
from helpers import orc, to_np

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ zero-length sequences in pack()
@pytest.mark.parametrize('lens', [[0, 3, 0, 2], [2, 0, 0, 5, 1, 0], [0, 0, 4], [3, 0]], ids=str)
@pytest.mark.parametrize('hidden,dtype', [((4,), torch.float32), ((16,), torch.float32), ((32,), torch.float32),
                                          ((40,), torch.float32), ((256,), torch.bfloat16), ((3,), torch.uint8)],
                         ids=['16B', '64B', '128B', '160B', '512B', '3B'])
def test_pack_with_empty_sequences(lens, hidden, dtype):
    """core/cast.py:41-49 handles empty sequences (they sort last and never show in batch_sizes): every valid
    sequence must arrive, whatever the row width (the generic mover and the (rank x time) tile kernel)."""
    g = torch.Generator().manual_seed(len(lens))
    n = sum(lens)
    data = (torch.randn((n,) + hidden, generator=g) * 4).to(dtype)
    lt = torch.tensor(lens, dtype=torch.long)
    srt = host_sort(lens)
    oc = orc.C(to_np(data), lt.numpy())
    op = orc.to_pack(oc, srt)
    c = ta.C(data.to(DEV), lt.to(DEV))
    for src, osrc in ((c, oc), (c.left(), orc.to_left(oc)), (c.right(), orc.to_right(oc))):
        p = src.pack()
        assert_same_seq(p, op, f'pack of {type(src).__name__}')
        # and back again: the oracle's P -> C is not defined with empty sequences in the reference (IndexError,
        # SURVEY §8a), here the round trip returns the original
        back = p.cat()
        assert torch.equal(back.data, c.data) and torch.equal(back.token_sizes.cpu(), lt)
        assert torch.equal(p.left().data, c.left().data)
        assert torch.equal(p.right().data, c.right().data)
        assert torch.equal(p.roll(1).cat().data, c.roll(1).data)
    if dtype in (torch.float32, torch.bfloat16):
        p = c.pack()
        f = data.float().numpy()
        for name in ('sum', 'mean', 'prod'):
            ref = getattr(orc, f'segment_{name}')(f, lt.numpy())
            out = getattr(ta, f'reduce_{name}')(p).float().cpu().numpy()
            np.testing.assert_allclose(out, ref, rtol=1e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-5)
        # max over an empty sequence: the reference's `initial` = the global minimum (reduce.py:35)
        ref = orc.segment_max(f, lt.numpy())
        out = ta.reduce_max(p).float().cpu().numpy()
        np.testing.assert_array_equal(out, ref)
        assert torch.equal(ta.reduce_max(p), ta.reduce_max(c))
        assert torch.equal(ta.reduce_logsumexp(p), ta.reduce_logsumexp(c))
        pf, of = ta.pack_reduce(c, 'sum', fused=True)
        assert torch.equal(pf.data, p.data) and torch.equal(of, ta.reduce_sum(p))


def test_pack_with_empty_sequences_metadata():
    lens = [0, 3, 0, 2]
    c = ta.C(torch.arange(5, dtype=torch.float32, device=DEV)[:, None].repeat(1, 64), torch.tensor(lens, device=DEV))
    p = c.pack()
    assert p.batch_sizes.tolist() == [2, 2, 1]
    assert sorted(p.sorted_indices.tolist()) == [0, 1, 2, 3] and p.sorted_indices.tolist()[:2] == [1, 3]
    assert p.size()[:2] == (2, 3)                         # the reference's P.size(): batch_sizes.max()
    assert ta.get_mask(p).tolist() == [[0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 0]]
    assert p.cat().token_sizes.tolist() == lens


# ------------------------------------------------------------------ scatter_* validation
def test_scatter_rejects_what_torch_rejects():
    t = torch.zeros(4, 8, device=DEV)
    idx = torch.tensor([0, 1, 1, 3], device=DEV)
    src = torch.ones(4, 8, device=DEV)
    ta.scatter_sum(t, idx, src)
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(t.half(), idx, src)                          # dtype mismatch (would overflow the half buffer)
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(t, idx, torch.ones(4, 9, device=DEV))        # row shape mismatch
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(t, idx[:3], src)                             # index shorter than source
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(t, torch.cat([idx, idx]), src)               # index longer than source
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(t, idx.view(2, 2), src)                      # index not 1-D
    with pytest.raises(ta.RuaError):
        ta.scatter_max(t, idx.float(), src)                         # index not integral


@pytest.mark.parametrize('name', ['sum', 'mean', 'max', 'min', 'prod', 'logsumexp'])
@pytest.mark.parametrize('include_self', [False, True])
def test_scatter_into_non_contiguous_tensor(name, include_self):
    """A dense but transposed `tensor`: the result is row-major data of the same values torch computes."""
    g = torch.Generator().manual_seed(3)
    t = torch.randn(6, 5, generator=g).t().to(DEV)            # [5, 6] view with strides (1, 5)
    assert not t.is_contiguous()
    idx = torch.tensor([4, 0, 0, 2, 4, 4, 1], device=DEV)
    src = torch.randn(7, 6, generator=g).to(DEV)
    out = getattr(ta, f'scatter_{name}')(t, idx, src, include_self=include_self)
    ref = getattr(orc, f'scatter_{name}')(t.cpu().contiguous().numpy(), idx.cpu().numpy(), src.cpu().numpy(), include_self)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    assert out.is_contiguous()


def test_scatter_int32_index():
    t = torch.zeros(3, 4, device=DEV)
    src = torch.arange(20, dtype=torch.float32, device=DEV).view(5, 4)
    idx = torch.tensor([2, 0, 2, 1, 0], device=DEV)
    assert torch.equal(ta.scatter_sum(t, idx.int(), src), ta.scatter_sum(t, idx, src))


# ------------------------------------------------------------------ inference mode
def test_hot_path_under_inference_mode():
    lens = [5, 2, 9, 1, 4]
    with torch.inference_mode():
        xs = [torch.randn(n, 32, device=DEV) for n in lens]
        c = ta.C.new(xs)
        p = c.pack()
        assert p.data.shape == (sum(lens), 32) and p.batch_sizes.tolist()[0] == 5
        assert c.size()[:2] == (5, 9)
        l = c.left()
        assert torch.equal(l.cat().data, c.data)
        s = ta.reduce_sum(p)
        ref = torch.stack([x.sum(0) for x in xs])
        torch.testing.assert_close(s, ref, rtol=1e-5, atol=1e-5)
        c2 = ta.with_host_sizes(c.data, torch.tensor(lens))
        assert torch.equal(c2.pack().data, p.data)
        c3 = ta.C(c.data, torch.tensor(lens, device=DEV))            # device-only lengths: D2H + memo
        assert torch.equal(c3.pack().data, p.data) and torch.equal(c3.roll(2).data, c.roll(2).data)
        assert torch.equal(ta.segment_max(c.data, c.token_sizes), torch.stack([x.max(0).values for x in xs]))


# ------------------------------------------------------------------ scatter_prod gradient where tensor == 0
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_scatter_prod_include_self_gradient_with_zeros_in_tensor(dtype):
    g = torch.Generator().manual_seed(11)
    t = (torch.rand(5, 6, generator=g, dtype=dtype) + 0.5)
    t[1, 2] = 0.0
    t[3] = 0.0
    t[4, 0] = 0.0                       # row 4 is named by no index: d out / d tensor = 1 there
    idx = torch.tensor([0, 1, 1, 3, 3, 3, 0])
    src = torch.rand(7, 6, generator=g, dtype=dtype) + 0.5
    src[4, 1] = 0.0
    w = torch.randn(5, 6, generator=g, dtype=dtype)

    def run(fn, dev):
        tt, ss = t.to(dev).detach().clone().requires_grad_(), src.to(dev).detach().clone().requires_grad_()
        out = fn(tt, idx.to(dev), ss)
        (out * w.to(dev)).sum().backward()
        return out.detach().cpu(), tt.grad.cpu(), ss.grad.cpu()

    ref = run(lambda a, i, s: torch.index_reduce(a, 0, i, s, 'prod', include_self=True), 'cpu')
    got = run(lambda a, i, s: ta.scatter_prod(a, i, s, include_self=True), DEV)
    for r, o, what in zip(ref, got, ('out', 'grad tensor', 'grad source')):
        assert not torch.isnan(o).any(), what
        torch.testing.assert_close(o, r, rtol=1e-5, atol=1e-6, msg=lambda m: f'{what}: {m}')


# ------------------------------------------------------------------ the host mirror is private
def test_with_host_sizes_keeps_its_own_copy():
    lens = torch.tensor([3, 1, 4, 2])
    data = torch.randn(10, 16, device=DEV)
    c = ta.with_host_sizes(data, lens)
    expect = ta.C(data, torch.tensor([3, 1, 4, 2], device=DEV)).pack()
    lens[:] = torch.tensor([1, 1, 1, 1])            # a loader recycling its buffer
    p = c.pack()
    assert p.batch_sizes.tolist() == [4, 3, 2, 1]
    assert torch.equal(p.data, expect.data) and torch.equal(p.sorted_indices, expect.sorted_indices)
    assert c.size()[:2] == (4, 4)


def test_with_host_sizes_survives_a_refill_that_bumps_no_version():
    """a loader that refills its buffer through a numpy view (or data_ptr) leaves `_version` alone; the
    mirror is the library's own copy, so later pack() / size() still see the lengths the container was built from."""
    buf = np.array([3, 1, 4, 2], dtype=np.int64)
    lens = torch.from_numpy(buf)
    data = torch.randn(10, 16, device=DEV)
    with torch.inference_mode():
        c_inf = ta.with_host_sizes(data, lens.clone())          # inference tensors keep no version at all
    c = ta.with_host_sizes(data, lens)
    v = lens._version
    buf[:] = [1, 1, 1, 1]
    assert lens._version == v and lens.tolist() == [1, 1, 1, 1]
    expect = ta.C(data, torch.tensor([3, 1, 4, 2], device=DEV)).pack()
    for z in (c, c_inf):
        p = z.pack()
        assert p.batch_sizes.tolist() == [4, 3, 2, 1] and torch.equal(p.data, expect.data)
        assert z.size()[:2] == (4, 4)


# ------------------------------------------------------------------ setitem: autograd, version counter, negative rows
def test_setitem_is_seen_by_autograd_and_by_the_version_counter():
    """`tensor[Z] = value` / `container[key] = value` went through the mover into raw.detach(): a gradient
    to `value` was dropped, a leaf requiring grad was overwritten silently, `_version` stayed.  Now a write autograd
    must see takes torch's own setitem (what the reference does: core/set.py:10-18); the others bump the version."""
    ta.patch_tensor_indexing()
    try:
        g = torch.Generator().manual_seed(3)
        lens = torch.tensor([3, 1, 4], device=DEV)
        base = torch.randn(8, 5, generator=g).to(DEV)
        rows = torch.tensor([6, 0, 3], device=DEV)
        key = ta.C(rows, torch.tensor([2, 1], device=DEV))
        # (1) gradient to `value` and through a non-leaf `self`
        value = torch.randn(3, 5, generator=g).to(DEV).requires_grad_(True)
        src = base.clone().requires_grad_(True)
        buf = src * 2.0
        buf[key] = value
        ref_v = value.detach().clone().requires_grad_(True)
        ref_s = base.clone().requires_grad_(True)
        ref = ref_s * 2.0
        ref[rows] = ref_v
        w = torch.randn(8, 5, generator=g).to(DEV)
        (buf * w).sum().backward()
        (ref * w).sum().backward()
        assert torch.equal(buf.detach(), ref.detach())
        assert torch.equal(value.grad, ref_v.grad) and torch.equal(src.grad, ref_s.grad)
        # (2) the container form, (batch_ptr, token_ptr) keys
        value2 = torch.randn(2, 5, generator=g).to(DEV).requires_grad_(True)
        c = ta.C(base.clone(), lens)
        c[torch.tensor([2, 0], device=DEV), torch.tensor([1, 2], device=DEV)] = value2
        (c.data * w).sum().backward()
        assert torch.equal(c.data[5].detach(), value2[0].detach()) and torch.equal(c.data[2].detach(), value2[1].detach())
        assert torch.equal(value2.grad, w[[5, 2]])
        # (3) a leaf that requires grad: torch's own error, not a silent overwrite
        leaf = base.clone().requires_grad_(True)
        with pytest.raises(RuntimeError):
            leaf[key] = 1.0
        # (4) no autograd involved: the mover writes, and the version counter moves
        plain = base.clone()
        v0 = plain._version
        plain[key] = 7.0
        assert plain._version > v0 and torch.equal(plain[rows], torch.full((3, 5), 7.0, device=DEV))
        c2 = ta.C(base.clone(), lens)
        v0 = c2.data._version
        c2[torch.tensor([1], device=DEV), torch.tensor([0], device=DEV)] = 9.0
        assert c2.data._version > v0 and bool((c2.data[3] == 9.0).all())
        with torch.no_grad():                       # a leaf under no_grad: allowed, as in torch
            leaf[key] = 1.0
        assert bool((leaf[rows] == 1.0).all())
    finally:
        ta.unpatch_tensor_indexing()


def test_negative_rows_wrap_like_torch():
    """Flat row keys (container[tensor], container[Z], tensor[Z], and their setitem twins) wrap negative entries the
    way torch's indexing does — the reference hands them to torch (core/get.py:29, core/set.py:30)."""
    ta.patch_tensor_indexing()
    try:
        data = torch.randn(9, 4, device=DEV)
        c = ta.C(data, torch.tensor([4, 5], device=DEV))
        rows = torch.tensor([-1, 0, -9, 3, -4], device=DEV)
        assert torch.equal(c[rows], data[rows])
        z = ta.C(rows, torch.tensor([2, 3], device=DEV))
        assert torch.equal(c[z].data, data[rows]) and torch.equal(data[z].data, data[rows])
        buf, ref = data.clone(), data.clone()
        buf[z] = 5.0
        ref[rows] = 5.0
        assert torch.equal(buf, ref)
        x = data.clone().requires_grad_(True)
        ta.C(x, c.token_sizes)[rows].sum().backward()
        xr = data.clone().requires_grad_(True)
        xr[rows].sum().backward()
        assert torch.equal(x.grad, xr.grad)
    finally:
        ta.unpatch_tensor_indexing()


# ------------------------------------------------------------------ row gathers: index dtypes, deterministic adjoint
def test_gather_with_narrow_int_and_bool_keys():
    data = torch.randn(12, 6, device=DEV)
    c = ta.C(data, torch.tensor([5, 3, 4], device=DEV))
    idx = torch.tensor([[3, 0], [11, 3]], device=DEV)
    ref = data[idx]
    for dt in (torch.long, torch.int32, torch.int16):
        assert torch.equal(c[idx.to(dt)], ref)
    mask = torch.zeros(12, dtype=torch.bool, device=DEV)
    mask[[1, 4, 9]] = True
    assert torch.equal(c[mask], data[mask])
    buf = data.clone()
    cc = ta.C(buf, c.token_sizes)
    cc[idx.int()] = 7.0
    ref2 = data.clone()
    ref2[idx] = 7.0
    assert torch.equal(buf, ref2)


@pytest.mark.parametrize('kind', ['C', 'L', 'P', 'R'])
def test_gather_backward_is_a_deterministic_scatter_sum(kind):
    """X[batch_ptr, token_ptr] with repeated pairs: the adjoint sums the repeats (torch: index_add_) — here through the
    bucketed reducer, bit-identical from run to run."""
    g = torch.Generator().manual_seed(7)
    lens = [4, 1, 6, 3]
    xs = [torch.randn(n, 5, generator=g) for n in lens]
    bp = torch.tensor([2, 0, 2, 2, 3, 0, 2], device=DEV)
    tp = torch.tensor([5, 3, 5, 0, 1, 3, 5], device=DEV)
    w = torch.randn(7, 5, generator=g).to(DEV)
    grads = []
    for _ in range(2):
        c = ta.C.new([x.to(DEV) for x in xs])
        z = {'C': c, 'L': c.left(), 'P': c.pack(), 'R': c.right()}[kind]
        leaf = z.data.detach().clone().requires_grad_()
        z = z._replace(data=leaf)
        (z[bp, tp] * w).sum().backward()
        grads.append(leaf.grad.clone())
    assert torch.equal(grads[0], grads[1])
    # expectation from the catted form: token (b, t) gets the sum of the weights of the pairs that name it
    exp = torch.zeros(sum(lens), 5, device=DEV)
    off = torch.tensor([0, 4, 5, 11], device=DEV)
    exp.index_add_(0, off[bp] + tp, w)
    back = {'C': lambda d: d, 'L': lambda d: ta.L(d, c.token_sizes).cat().data, 'R': lambda d: ta.R(d, c.token_sizes).cat().data,
            'P': lambda d: c.pack()._replace(data=d).cat().data}[kind](grads[0])
    torch.testing.assert_close(back, exp, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ long runs of zero-length sequences
@pytest.mark.parametrize('hidden', [(8,), (256,), (1,)], ids=['32B', '1KiB', '4B'])
def test_runs_of_empty_sequences_between_real_ones(hidden):
    """More than 64 sequence starts inside one wave's rows: the wave-cooperative row resolution (a 64-entry window of
    offsets) cannot cover them and every lane must fall back to its own search — for the mover (CAT destinations), for
    ptr()/idx() and for the reducers."""
    g = torch.Generator().manual_seed(3)
    lens = []
    for k in range(40):
        lens += [int(torch.randint(1, 9, (1,), generator=g))] + [0] * int(torch.randint(0, 300, (1,), generator=g))
    lens += [5, 0, 0, 3]
    lt = torch.tensor(lens, dtype=torch.long)
    n = int(lt.sum())
    data = torch.randn((n,) + hidden, generator=g)
    c = ta.C(data.to(DEV), lt.to(DEV))
    oc = orc.C(data.numpy(), lt.numpy())
    # enumeration
    bp, tp = c.ptr()
    obp, otp = orc.ptr(oc)
    assert np.array_equal(to_np(bp), obp) and np.array_equal(to_np(tp), otp)
    # pads and back (CAT destination = the fallback path), roll inside C
    left, right = c.left(-2.0), c.right(-2.0)
    assert np.array_equal(to_np(left.data), orc.to_left(oc, -2.0).data)
    assert np.array_equal(to_np(right.data), orc.to_right(oc, -2.0).data)
    assert torch.equal(left.cat().data, c.data) and torch.equal(right.cat().data, c.data)
    assert np.array_equal(to_np(c.roll(2).data), orc.roll(oc, 2).data)
    lidx = left.idx()
    assert np.array_equal(to_np(lidx.data), orc.idx(orc.to_left(oc, -2.0)).data)
    # pack and back
    p = c.pack()
    assert_same_seq(p, orc.to_pack(oc, host_sort(lens)), 'pack')
    assert torch.equal(p.cat().data, c.data) and p.cat().token_sizes.tolist() == lens
    # reductions: empty segments take the reference's initial
    for name in ('sum', 'max', 'logsumexp', 'mean'):
        ref = getattr(orc, f'segment_{name}')(data.numpy(), lt.numpy())
        got = getattr(ta, f'segment_{name}')(c.data, c.token_sizes).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5, err_msg=name)
        got_p = getattr(ta, f'reduce_{name}')(p).cpu().numpy()
        np.testing.assert_allclose(got_p, ref, rtol=1e-5, atol=1e-5, err_msg=name + ' over P')


def test_empty_segments_at_a_grid_that_hits_the_cap():
    """max / min / logsumexp with empty segments among 300 000: the one trailing launch walks the payload for the
    global extreme with its (capped) grid, the workgroup that finishes last patches the empty rows — against the oracle,
    twice in a row (the persistent scratch must come back zeroed)."""
    g = torch.Generator().manual_seed(12)
    lens = torch.randint(0, 4, (300_000,), generator=g)
    data = torch.randn(int(lens.sum()), 8, generator=g)
    d, l = data.to(DEV), lens.to(DEV)
    for _ in range(2):
        for name in ('max', 'min', 'logsumexp'):
            ref = getattr(orc, f'segment_{name}')(data.numpy(), lens.numpy())
            got = getattr(ta, f'segment_{name}')(d, l).cpu().numpy()
            if name == 'logsumexp':
                np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
            else:
                np.testing.assert_array_equal(got, ref)
        # and a call with nothing empty and no NaN right after: untouched by leftovers
        full = torch.ones(300_000, dtype=torch.long, device=DEV)
        assert torch.equal(ta.segment_max(d[:300_000], full), d[:300_000])
    # a NaN poisons every segment (reduce.py:35: initial = tensor.min() is NaN), then the next call is clean again
    d2 = d.clone()
    d2[12345, 3] = float('nan')
    assert torch.isnan(ta.segment_max(d2, l)).all()
    assert not torch.isnan(ta.segment_max(d, l)).any()


def test_pack_reduce_picks_the_two_kernel_form_for_few_sequences():
    """The fused kernel gives one wave to each sequence; with a few thousand sequences the automatic choice is pack()
    followed by the team reducer (same PackedSequence, sums in the same fp32 arithmetic); fused=True still forces
    the one-pass kernel."""
    g = torch.Generator().manual_seed(4)
    lens = torch.randint(100, 400, (512,), generator=g)
    data = torch.randn(int(lens.sum()), 64, generator=g).to(DEV)
    c = ta.with_host_sizes(data, lens)
    p_auto, s_auto = ta.pack_reduce(c, 'sum')
    p_two = c.pack()
    assert torch.equal(p_auto.data, p_two.data) and torch.equal(s_auto, ta.reduce_sum(p_two))
    p_fused, s_fused = ta.pack_reduce(c, 'sum', fused=True)
    assert torch.equal(p_fused.data, p_two.data)
    torch.testing.assert_close(s_fused, s_auto, rtol=1e-5, atol=1e-4)
    ref = torch.stack([x.sum(0) for x in torch.split(data.double(), lens.tolist())])
    assert ((s_auto.double() - ref).abs() <= 1e-5 * torch.stack([x.abs().sum(0) for x in torch.split(data.double(), lens.tolist())])).all()


@pytest.mark.parametrize('name', ['sum', 'mean', 'max', 'prod', 'logsumexp'])
@pytest.mark.parametrize('dim', [1, -1, 2])
def test_scatter_along_another_dim(name, dim):
    """reduce.py:6-31 hand `dim` to torch.index_reduce / index_add; here any dim is brought to the front and back."""
    g = torch.Generator().manual_seed(21)
    t = torch.rand(3, 5, 4, generator=g) + 0.5
    n_src = 7
    shape = list(t.shape)
    shape[dim] = n_src
    src = torch.rand(shape, generator=g) + 0.5
    idx = torch.randint(0, t.shape[dim], (n_src,), generator=g)
    for inc in (False, True):
        td, sd = t.to(DEV).requires_grad_(), src.to(DEV).requires_grad_()
        out = getattr(ta, f'scatter_{name}')(td, idx.to(DEV), sd, include_self=inc, dim=dim)
        tc, sc = t.clone().requires_grad_(), src.clone().requires_grad_()
        if name == 'sum' and not inc:
            ref = torch.zeros_like(tc).index_add(dim, idx, sc)
        elif name == 'sum':
            ref = tc.index_add(dim, idx, sc)
        elif name == 'logsumexp':
            d = dim % 3
            ref = getattr(orc, 'scatter_logsumexp')(t.movedim(d, 0).contiguous().numpy(), idx.numpy(), src.movedim(d, 0).contiguous().numpy(), inc)
            ref = torch.from_numpy(ref).movedim(0, d)
        else:
            ref = torch.index_reduce(tc, dim, idx, sc, {'mean': 'mean', 'max': 'amax', 'prod': 'prod'}[name], include_self=inc)
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        if name != 'logsumexp':
            w = torch.randn(out.shape, generator=g)
            (out * w.to(DEV)).sum().backward()
            (ref * w).sum().backward()
            torch.testing.assert_close(sd.grad.cpu(), sc.grad, rtol=1e-5, atol=1e-6)
            if tc.grad is not None and td.grad is not None:
                torch.testing.assert_close(td.grad.cpu(), tc.grad, rtol=1e-5, atol=1e-6)


def test_two_host_threads_share_one_stream():
    """Two Python threads enqueueing on the same stream (ctypes drops the GIL around every launch): the reduce and its
    trailing rua_fill_empty share one scratch and must stay back to back; pack() with device-only lengths must never be
    handed a staging slot another thread is filling."""
    import threading
    g = torch.Generator().manual_seed(33)
    cases = []
    for k in range(2):
        lens = torch.randint(0, 9, (1500 + 700 * k,), generator=g)          # a third of the segments empty-ish
        lens[::3] = 0
        data = torch.randn(int(lens.sum()), 16, generator=g)
        pl = torch.randint(1, 40, (1200 + 4900 * k,), generator=g)         # >= 1 024 lengths: the side-stream upload;
                                                                            # the second case >= 4 096: the sort on the helper thread
        pd = torch.randn(int(pl.sum()), 8, generator=g)
        exp_max = orc.segment_max(data.numpy(), lens.numpy())
        exp_pack = orc.to_pack(orc.C(pd.numpy(), pl.numpy()), host_sort(pl))
        cases.append((data.to(DEV), lens.to(DEV), exp_max, pd.to(DEV), pl, exp_pack))
    torch.cuda.synchronize()
    errors = []

    def work(k):
        try:
            data, lens, exp_max, pd, pl, exp_pack = cases[k]
            for _ in range(150):
                got = ta.segment_max(data, lens)
                p = ta.C(pd, pl.to(DEV)).pack()                               # device-only lengths: read back, sort, upload
                assert np.array_equal(got.cpu().numpy(), exp_max), 'segment_max with empty segments'
                assert np.array_equal(p.data.cpu().numpy(), exp_pack.data), 'pack payload'
                assert np.array_equal(p.sorted_indices.cpu().numpy(), exp_pack.sorted_indices), 'sorted_indices'
        except BaseException as e:      # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_pack_with_device_only_lengths_overlaps_the_host_sort():
    """C(data, token_sizes_on_device).pack() with >= 4 096 sequences sorts on the library's helper thread while the
    calling thread derives batch_sizes and both offset scans on the host (core._pack_meta_overlapped): the same
    PackedSequence, bit for bit, as with host-known lengths and as the oracle's; ties beyond the 16-element leaf; zero
    lengths; a second pack() of the same container is served from the memo; the next op (P -> C, reduce) sees the
    metadata the overlapped path uploaded."""
    g = torch.Generator().manual_seed(77)
    for B, hi in ((4096, 3), (5000, 40), (70001, 9)):
        lens = torch.randint(0, hi + 1, (B,), generator=g)
        lens[B // 2] = hi
        N = int(lens.sum())
        data = torch.randn(N, 4, generator=g)
        want = orc.to_pack(orc.C(data.numpy(), lens.numpy()), host_sort(lens))
        c = ta.C(data.to(DEV), lens.to(DEV))
        p = c.pack()
        assert_same_seq(p, want, f'pack B={B}')
        assert c.pack().data is not p.data and torch.equal(c.pack().data, p.data)
        assert torch.equal(c.pack().sorted_indices, p.sorted_indices)            # memo: the same tensors
        ref = ta.with_host_sizes(data.to(DEV), lens).pack()
        assert torch.equal(ref.data, p.data) and torch.equal(ref.unsorted_indices, p.unsorted_indices)
        assert torch.equal(ref.batch_sizes, p.batch_sizes)
        assert torch.equal(p.cat().data.cpu(), data)
        np.testing.assert_allclose(ta.reduce_sum(p).cpu().numpy(), orc.segment_sum(data.numpy(), lens.numpy()), rtol=1e-5, atol=1e-5)
        # a container whose lengths are a strided view (not contiguous int64): normalised first, offsets not memoised on it
        wide = torch.stack([lens, lens], 1).to(DEV)[:, 0]
        assert_same_seq(ta.C(data.to(DEV), wide).pack(), want, f'pack strided lens B={B}')
        narrow = ta.C(data.to(DEV), lens.to(DEV).int()).pack()           # int32 lengths on the device
        assert torch.equal(narrow.data, p.data) and torch.equal(narrow.sorted_indices, p.sorted_indices)
        assert torch.equal(narrow.batch_sizes, p.batch_sizes)


def test_a_hot_bucket_is_split_not_streamed_by_one_wave():
    """scatter_* on a skewed histogram (a third of the entries in ONE of 100 000 buckets, average bucket far below the
    256 rows that used to arm the split): the hot bucket must be cut into parts — one wave streaming it took 40 ms
    here (171 ms at the north-star row count) against well under a millisecond — for floats and for integers, with
    the same values as torch's own index_add_ / bincount."""
    import time
    g = torch.Generator().manual_seed(5)
    S, Mn, H = 100000, 1 << 22, 64
    idx = torch.where(torch.rand(Mn, generator=g) < 0.33, torch.tensor(5), torch.randint(0, S, (Mn,), generator=g)).to(DEV)
    src = torch.randn(Mn, H, generator=g).to(DEV)
    ten = torch.zeros(S, H, device=DEV)
    ones = torch.ones(Mn, dtype=torch.long, device=DEV)
    zeros = torch.zeros(S, dtype=torch.long, device=DEV)
    got = ta.scatter_sum(ten, idx, src)
    cnt = ta.scatter_sum(zeros, idx, ones)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = ta.scatter_sum(ten, idx, src)
    cnt = ta.scatter_sum(zeros, idx, ones)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.equal(cnt, torch.bincount(idx, minlength=S))
    want = torch.zeros(S, H, device=DEV, dtype=torch.float64).index_add_(0, idx, src.double())
    scale = torch.zeros(S, H, device=DEV, dtype=torch.float64).index_add_(0, idx, src.double().abs())
    assert bool(((got.double() - want).abs() <= 1e-5 * scale + 1e-6).all())
    assert dt < 0.015, f'{dt * 1e3:.1f} ms: the hot bucket was streamed by one wave'
