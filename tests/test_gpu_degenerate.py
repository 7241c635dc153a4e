"""Degenerate length distributions, on the GPU: one giant sequence among thousands of short ones (thousands of time
chunks for the (rank x time) tile kernels, a PackedSequence whose tail is one row per step), every index of a scatter
naming ONE bucket.  Values against plain torch restatements of the reference's maps (core/view.py:47-58 pack order,
core/cast.py:8-38, select/roll.py:26-30, reduce.py:6-61)."""
import time

import pytest
import torch

import torchrua_amd as ta
from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _giant(H, dtype, seed, few_long=False):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, 17, (3000,), generator=g)
    lens[1234] = 200_000
    lens[7] = 0
    if few_long:        # twenty sequences of ~9 000 steps: more than 64 time chunks of FULL tiles (the tile kernels' own
        lens = torch.cat([torch.randint(1, 17, (2000,), generator=g), torch.randint(8000, 10000, (20,), generator=g)])   # search)
    N = int(lens.sum())
    data = torch.randn(N, H, generator=g).to(dtype)
    return lens, data


def _packed_rows(lens, p):
    """Row of token (b, t) in the PackedSequence: boff[t] + rank[b] (core/view.py:47-58)."""
    B = lens.numel()
    off = torch.cumsum(lens, 0) - lens
    bsz = p.batch_sizes.cpu()
    boff = torch.cumsum(bsz, 0) - bsz
    rank = p.unsorted_indices.cpu()
    b = torch.repeat_interleave(torch.arange(B), lens)
    t = torch.arange(int(lens.sum())) - off[b]
    return boff[t] + rank[b], b, t


@pytest.mark.parametrize('few_long', [False, True])
@pytest.mark.parametrize('H,dtype', [(8, torch.bfloat16), (16, torch.bfloat16), (32, torch.float32), (512, torch.bfloat16)])
def test_one_giant_sequence_among_short_ones(H, dtype, few_long):
    """few_long=False: the tiles would be mostly dead cells, the generic mover takes the narrow rows too
    (_meta.TILE_MIN_LIVE_INV); few_long=True: thousands of live tiles over more than 64 time chunks."""
    lens, data = _giant(H, dtype, H, few_long)
    big = int(lens.argmax())
    c = ta.with_host_sizes(data.to(DEV), lens)
    p = c.pack()
    si = p.sorted_indices.cpu()
    assert torch.equal(si, torch.sort(lens, descending=True)[1])
    assert torch.equal(p.unsorted_indices.cpu()[si], torch.arange(lens.numel()))
    rows, b, t = _packed_rows(lens, p)
    want = torch.empty_like(data)
    want[rows] = data
    assert torch.equal(p.data.cpu(), want), 'pack'
    assert torch.equal(p.cat().data.cpu(), data), 'P.cat'
    # roll by one inside the PackedSequence (select/roll.py:26-30): token t takes the value of token (t - 1) mod len
    rolled = p.roll(1)
    src_t = (t - 1) % lens[b]
    off = torch.cumsum(lens, 0) - lens
    want_roll = torch.empty_like(data)
    want_roll[rows] = data[off[b] + src_t]
    assert torch.equal(rolled.data.cpu(), want_roll), 'P.roll'
    # the reductions: the giant sequence through the split (the host knows the lengths) and without it (device lengths)
    ref = torch.zeros(lens.numel(), H, dtype=torch.float64).index_add_(0, b, data.double())
    mag = torch.zeros(lens.numel(), H, dtype=torch.float64).index_add_(0, b, data.double().abs())
    tol = (1e-5 if dtype == torch.float32 else 1e-2) * mag + 1e-6
    for name, got in (('reduce_sum(p)', ta.reduce_sum(p)), ('reduce_sum(c)', ta.reduce_sum(c)),
                      ('segment_sum, device lengths', ta.segment_sum(data.to(DEV), lens.to(DEV)))):
        assert bool(((got.double().cpu() - ref).abs() <= tol).all()), name
    mx = ta.reduce_max(p).cpu()
    assert torch.equal(mx[big], data[off[big]:off[big] + lens[big]].max(0)[0]), 'max of the longest sequence'


def test_every_index_names_one_bucket():
    """index_buckets / scatter_* when the histogram is ONE spike among 65 536 destinations (the most-significant-digit
    first bucket builder puts every entry in one bin, the reducers cut the bucket by position / into parts)."""
    Mn, S, H = 1 << 22, 65536, 16
    g = torch.Generator().manual_seed(2)
    idx = torch.full((Mn,), 4242, device=DEV)
    src = torch.randn(Mn, H, generator=g).to(DEV)
    ones = torch.ones(Mn, dtype=torch.int32, device=DEV)
    ta.scatter_sum(torch.zeros(S, H, device=DEV), idx, src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = ta.scatter_sum(torch.zeros(S, H, device=DEV), idx, src)
    cnt = ta.scatter_sum(torch.zeros(S, dtype=torch.int32, device=DEV), idx, ones)
    hi = ta.scatter_max(torch.zeros(S, H, device=DEV), idx, src)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(cnt[4242]) == Mn and int(cnt.sum()) == Mn
    want = src.double().sum(0)
    assert bool(((got[4242].double() - want).abs() <= 1e-5 * src.double().abs().sum(0)).all())
    assert int((got != 0).any(1).sum()) == 1
    assert torch.equal(hi[4242], src.max(0)[0])
    assert dt < 0.05, f'{dt * 1e3:.1f} ms for three scatters of 4 M entries into one bucket'


@pytest.mark.parametrize('giant', [False, True])
@pytest.mark.parametrize('H,dtype', [(4, torch.float32), (8, torch.bfloat16), (8, torch.float32), (16, torch.bfloat16), (2, torch.float64)])
def test_four_sequences_per_wave_at_narrow_rows(H, dtype, giant):
    """Rows of <= 32 bytes, a few hundred rows per sequence: four sequences share a wave, 16 / 8 rows of each per
    instruction (glog > 0 in make_unit), whether the host knows the lengths or not — every wave checks its own four
    lengths and walks them one after the other when they are far apart (giant=True: one sequence of 50 000 rows among
    them; with host-known lengths that one arms the long-sequence split instead).  Against the oracle's sequential
    folds; gradients against the one-row-at-a-time backward of the same library in float64."""
    import numpy as np
    from helpers import orc
    g = torch.Generator().manual_seed(100 + H)
    B = 40_000
    lens = torch.randint(50, 301, (B,), generator=g)
    lens[5], lens[6], lens[B - 1] = 0, 1, 700
    if giant:
        lens[7777] = 50_000
    N = int(lens.sum())
    data = (torch.randn(N, H, generator=g) * 0.5).to(dtype)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    dd = data.to(DEV)
    ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    for z in (ta.with_host_sizes(dd, lens), ta.C(dd, lens.to(DEV))):
        for name in ('sum', 'mean', 'max', 'min', 'logsumexp'):
            ref = getattr(orc, f'segment_{name}')(f, lens.numpy()).astype(np.float64)
            got = getattr(ta, f'reduce_{name}')(z).double().cpu().numpy()
            scale = float(np.abs(f).max()) * (int(lens.max()) if name == 'sum' else 1)
            np.testing.assert_allclose(got, ref, rtol=2e-5 + ulp, atol=2e-5 * scale + ulp + 1e-6, err_msg=name)
    if dtype in (torch.float32, torch.float64):
        tied = torch.randint(0, 3, (N, H), generator=g).to(dtype)
        cot = torch.randn(B, H, generator=g).to(dtype)
        for name in ('max', 'sum'):
            r = tied.clone().requires_grad_(True)
            torch.segment_reduce(r, name, lengths=lens, unsafe=True).backward(cot)
            for z_of in (lambda x: ta.with_host_sizes(x, lens), lambda x: ta.C(x, lens.to(DEV))):
                x = tied.clone().to(DEV).requires_grad_(True)
                getattr(ta, f'reduce_{name}')(z_of(x)).backward(cot.to(DEV))
                assert torch.allclose(x.grad.cpu(), r.grad, rtol=1e-5, atol=1e-6), f'grad {name}'


@pytest.mark.parametrize('H,dtype', [(4, torch.float32), (8, torch.float32), (16, torch.bfloat16), (32, torch.float32),
                                     (50, torch.float32), (100, torch.bfloat16), (128, torch.float32)])
def test_short_sequences_side_by_side(H, dtype):
    """Hundreds of thousands of short sequences whose lengths the host knows: adjacent sequences share a wave
    (RUA_OP_SHORT_SEQS).  Same values as the one-wave-per-sequence walk (device-only lengths): max / min bit for bit,
    empty sequences and the reference's `initial` included; the sums to rounding."""
    from torchrua_amd import _ops as O
    from torchrua_amd.layout import describe
    g = torch.Generator().manual_seed(H)
    B = 300_000
    lens = torch.randint(0, 9, (B,), generator=g)
    lens[::5003] = 40
    N = int(lens.sum())
    data = (torch.randn(N, H, generator=g) * 0.5).to(dtype).to(DEV)
    host = ta.with_host_sizes(data, lens)
    dev = ta.C(data, lens.to(DEV))
    assert O.short_seqs_hint(describe(host), H * data.element_size()) == ta._lib.OP_SHORT_SEQS
    assert O.short_seqs_hint(describe(dev), H * data.element_size()) == 0                  # lengths on the device only
    for name in ('sum', 'mean', 'max', 'min', 'prod', 'logsumexp'):
        fn = getattr(ta, f'reduce_{name}')
        a, b = fn(host), fn(dev)
        if name in ('max', 'min'):
            assert torch.equal(a, b), name
        else:
            tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
            assert torch.allclose(a.float(), b.float(), rtol=tol, atol=tol, equal_nan=True), name
    # with gradients: the forward of max / min also counts the ties (side by side too); same gradients either way
    if dtype == torch.float32:
        tied = torch.randint(0, 3, (N, H), generator=g).float().to(DEV)
        cot = torch.randn(B, H, generator=g).to(DEV)
        for name in ('max', 'sum', 'logsumexp'):
            grads = []
            for lens_arg in (lens, lens.to(DEV)):
                x = tied.clone().requires_grad_(True)
                z = ta.with_host_sizes(x, lens_arg) if not lens_arg.is_cuda else ta.C(x, lens_arg)
                getattr(ta, f'reduce_{name}')(z).backward(cot)
                grads.append(x.grad)
            assert torch.allclose(grads[0], grads[1], rtol=1e-5, atol=1e-6), f'grad {name}'
    # one long sequence among them: the hint is withdrawn (the wave would walk it with one lane group)
    lens2 = lens.clone()
    lens2[77] = 10_000
    data2 = torch.zeros(int(lens2.sum()), H, dtype=dtype, device=DEV)
    assert O.short_seqs_hint(describe(ta.with_host_sizes(data2, lens2)), H * data.element_size()) == 0
    # [r5] ... but not the side-by-side form: with nobody's word about the longest sequence every wave checks its own
    # lengths and walks them one after the other when they are far apart.  Device-only lengths and host-known lengths
    # without the hint, the 10 000-row sequence among them, against the oracle's sequential folds
    import numpy as np
    from helpers import orc
    d2 = (torch.randn(int(lens2.sum()), H, generator=g) * 0.5).to(dtype)
    f = d2.float().numpy()
    ulp = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    for z in (ta.C(d2.to(DEV), lens2.to(DEV)), ta.with_host_sizes(d2.to(DEV), lens2)):
        for name in ('sum', 'max', 'logsumexp'):
            ref = getattr(orc, f'segment_{name}')(f, lens2.numpy()).astype(np.float64)
            got = getattr(ta, f'reduce_{name}')(z).double().cpu().numpy()
            scale = float(np.abs(f).max()) * (10_000 if name == 'sum' else 1)
            np.testing.assert_allclose(got, ref, rtol=2e-5 + ulp, atol=2e-5 * scale + ulp + 1e-6, err_msg=name)


@pytest.mark.parametrize('H,dtype,skip', [(8, torch.bfloat16, 1), (16, torch.bfloat16, 3), (8, torch.float32, 2), (16, torch.float32, 1)])
def test_narrow_p_cat_into_a_view_that_starts_inside_a_line(H, dtype, skip):
    """P.cat at 16 / 32 / 64-byte rows gives every rank its own time origin so that its stores are whole 128-byte lines
    (rua_move.hip: TileTables); the origin depends on where the OUTPUT starts inside a line.  Write into a view that
    begins `skip` rows into a fresh allocation: same rows as the plain call, nothing outside the view touched."""
    from torchrua_amd import _ops as O
    from torchrua_amd.layout import describe
    g = torch.Generator().manual_seed(H + skip)
    lens = torch.randint(1, 200, (3000,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, generator=g).to(dtype).to(DEV)
    c = ta.with_host_sizes(data, lens)
    p = c.pack()
    buf = torch.full((N + 8, H), 7.0, dtype=dtype, device=DEV)
    out = buf[skip:skip + N]
    O.launch_move(O.MovePlan(describe(c), describe(p), data.shape), p.data, out=out)
    assert torch.equal(out, data)
    assert bool((buf[:skip] == 7).all()) and bool((buf[skip + N:] == 7).all())
    # and the other direction out of such a view (plain windows there: the reads tolerate the misalignment)
    src = buf[skip:skip + N]
    packed = torch.empty_like(data)
    O.launch_move(O.MovePlan(describe(p), describe(c), data.shape), src, out=packed)
    assert torch.equal(packed, p.data)


@pytest.mark.parametrize('H,dtype', [(64, torch.float32), (16, torch.bfloat16), (3, torch.float64)])
def test_mostly_empty_batch_under_max_min_logsumexp(H, dtype):
    """Nine sequences in ten are empty: their rows take the reference's `initial` — the global minimum / maximum of the
    payload (reduce.py:35,40,57), which the reduce tracks itself; the trailing launch patches them with every workgroup,
    whether or not the host knows the lengths (rounds 2-4: one workgroup, after a second walk over the payload).  For a
    CattedSequence and a PackedSequence, against the oracle."""
    import numpy as np
    from helpers import orc
    g = torch.Generator().manual_seed(7 + H)
    B = 150_000
    lens = torch.where(torch.rand(B, generator=g) < 0.9, torch.tensor(0), torch.randint(1, 40, (B,), generator=g))
    N = int(lens.sum())
    data = (torch.randn(N, H, generator=g) * 0.5).to(dtype)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    dd = data.to(DEV)
    host = ta.with_host_sizes(dd, lens)
    ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    for z in (host, ta.C(dd, lens.to(DEV)), host.pack()):
        for name in ('max', 'min', 'logsumexp'):
            ref = getattr(orc, f'segment_{name}')(f, lens.numpy()).astype(np.float64)
            got = getattr(ta, f'reduce_{name}')(z).double().cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-5 + ulp, atol=2e-5 + ulp, err_msg=name)
    # and again (the persistent scratch must have come back zeroed from either form)
    for z in (host, host.pack()):
        ref = orc.segment_max(f, lens.numpy()).astype(np.float64)
        np.testing.assert_allclose(ta.reduce_max(z).double().cpu().numpy(), ref, rtol=2e-5 + ulp, atol=2e-5 + ulp)


@pytest.mark.parametrize('H,dtype', [(8, torch.bfloat16), (8, torch.float32), (4, torch.float32)])
def test_four_sequences_per_wave_with_the_split_armed(H, dtype):
    """Lengths on the device only and 300 rows per sequence on average: nobody vouches for the longest sequence, so the
    long-sequence split is armed (`_meta.reduce_split_rows`) — up to round 4 that sent rows of <= 32 bytes back to one wave
    per sequence.  Now the four-per-wave kernel cuts long sequences itself (seg_reduce_ranks_kernel<SPLIT>: part 0 in
    place, the rest through the work list of the tail and combine kernels): a 700 000-row sequence, a 9 000-row one, empty
    ones and ordinary ones in the same launch, against the oracle's sequential folds."""
    import numpy as np
    from helpers import orc
    from torchrua_amd import _meta as M
    from torchrua_amd.layout import describe
    g = torch.Generator().manual_seed(300 + H)
    B = 36_000
    lens = torch.randint(200, 401, (B,), generator=g)
    lens[3], lens[4], lens[B - 1], lens[B - 2] = 0, 1, 0, 9_000
    lens[12_345] = 700_000
    N = int(lens.sum())
    data = (torch.randn(N, H, generator=g) * 0.5).to(dtype)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    dd = data.to(DEV)
    z = ta.C(dd, lens.to(DEV))
    assert M.reduce_split_rows(describe(z), H * dtype.itemsize) > 0, 'the split is expected to be armed here'
    ulp = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    for name in ('sum', 'max', 'min', 'logsumexp'):
        ref = getattr(orc, f'segment_{name}')(f, lens.numpy()).astype(np.float64)
        for _ in range(2):          # (twice: the scratch and the work list come back clean)
            got = getattr(ta, f'reduce_{name}')(z).double().cpu().numpy()
            scale = float(np.abs(f).max()) * (int(lens.max()) if name == 'sum' else 1)
            np.testing.assert_allclose(got, ref, rtol=2e-5 + ulp, atol=2e-5 * scale + ulp + 1e-6, err_msg=name)
