"""The reference's global `initial` of max / min / logsumexp (reduce.py:35,40,57: tensor.min() / tensor.max(), a second
pass over the payload plus a sync there) is tracked by the reduce ITSELF since round 5: every wave folds the opposite
extreme of the rows it reads and hands it to the scratch, the trailing launch writes it into the rows of empty segments
— or NaN everywhere when the payload holds a NaN — with every workgroup.  Every forward kernel is reached here (one
wave per sequence, two per workgroup, teams of waves, sequences side by side at narrow rows, column chunks, rows of
8 mod 16 bytes, the scalar path), with empty segments in different places, against the oracle; and the persistent
scratch must come back zeroed."""
import numpy as np
import pytest
import torch

import torchrua_amd as ta
from torchrua_amd import _ops as O
from gpu_util import DEV
from helpers import orc

pytestmark = pytest.mark.gpu


def _case(B, lo, hi, H, dtype, seed, empty_every=5):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    if empty_every:
        lens[seed % empty_every::empty_every] = 0
        lens[B - 1] = 0
    N = int(lens.sum())
    data = (torch.randn(N, H, generator=g) * 2).to(dtype)
    return lens, data


def _scratch_is_clean():
    torch.cuda.synchronize()
    return all(int(buf.abs().sum()) == 0 for buf in O._scratch.values())


# (B, lo, hi, H, dtype): which forward kernel the launcher picks for it
SHAPES = [
    (700, 1, 40, 512, torch.bfloat16),      # C: two waves per workgroup (rows >= 512 B); P: one wave per sequence
    (300, 200, 400, 64, torch.float32),     # few but long units: a team of waves per sequence
    (5000, 1, 30, 8, torch.bfloat16),       # 16-byte rows: sequences side by side in a wave (C: four per wave; P: adjacent ranks)
    (2000, 1, 12, 3, torch.float64),        # scalar path (rows of 24 bytes, one element per lane)
    (900, 1, 50, 500, torch.bfloat16),      # rows of 8 (mod 16) bytes: overlapping last lane
    (64, 1, 9, 1100, torch.float32),        # rows wider than one wave instruction (column chunks)
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: f'B{s[0]}_H{s[3]}_{str(s[4]).split(".")[-1]}')
def test_empty_segments_take_the_global_extreme(shape):
    B, lo, hi, H, dtype = shape
    lens, data = _case(B, lo, hi, H, dtype, seed=B + H)
    f = data.double().numpy() if dtype == torch.float64 else data.float().numpy()
    dd = data.to(DEV)
    host = ta.with_host_sizes(dd, lens)
    ulp = {torch.float32: 0.0, torch.float64: 0.0, torch.bfloat16: 2.0 ** -8}[dtype]
    for zi, z in enumerate((host, ta.C(dd, lens.to(DEV)), host.pack())):
        for name in ('max', 'min', 'logsumexp'):
            ref = getattr(orc, f'segment_{name}')(f, lens.numpy()).astype(np.float64)
            for _ in range(2):                    # (twice: the scratch came back clean)
                out = getattr(ta, f'reduce_{name}')(z)
                np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=2e-5 + ulp, atol=2e-5 + ulp,
                                           err_msg=f'{name} layout {zi}')
    assert _scratch_is_clean()


def test_a_nan_poisons_every_segment():
    """A NaN anywhere makes the reference's `initial` NaN, and torch.segment_reduce folds it into EVERY segment."""
    lens, data = _case(400, 1, 20, 96, torch.float32, seed=3, empty_every=0)
    data[int(lens[:200].sum()) + 1, 17] = float('nan')
    dd = data.to(DEV)
    for z in (ta.C(dd, lens.to(DEV)), ta.with_host_sizes(dd, lens).pack()):
        for name in ('max', 'min', 'logsumexp'):
            out = getattr(ta, f'reduce_{name}')(z)
            assert bool(torch.isnan(out).all()), name
    assert _scratch_is_clean()
    # and the next call on clean data is clean again
    lens2, data2 = _case(400, 1, 20, 96, torch.float32, seed=4)
    out = ta.segment_max(data2.to(DEV), lens2.to(DEV))
    np.testing.assert_array_equal(out.cpu().numpy(), orc.segment_max(data2.numpy(), lens2.numpy()))


def test_the_extreme_sits_in_a_split_sequence():
    """The global minimum lives in the LAST part of a sequence that is cut into parts (the tail kernel's waves hand their
    share of the extreme over too), and an empty segment takes it."""
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(1, 9, (3000,), generator=g)
    lens[1000], lens[17], lens[2999] = 300_000, 0, 0
    N = int(lens.sum())
    data = torch.randn(N, 16, generator=g)
    row = int(lens[:1000].sum()) + 299_990
    data[row, 5] = -77.0
    dd = data.to(DEV)
    for z in (ta.with_host_sizes(dd, lens), ta.C(dd, lens.to(DEV))):
        out = ta.reduce_max(z).cpu().numpy()
        assert out[17, 0] == -77.0 and out[2999, 9] == -77.0
        np.testing.assert_array_equal(out, orc.segment_max(data.numpy(), lens.numpy()))
    assert _scratch_is_clean()


def test_the_differentiable_forward_tracks_it_too():
    """reduce_max under autograd runs the tie-counting forward (RUA_MAX_T); empty segments get the global minimum and a
    zero gradient."""
    lens, data = _case(500, 1, 30, 64, torch.float32, seed=11)
    x = data.to(DEV).requires_grad_(True)
    out = ta.segment_max(x, lens.to(DEV))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), orc.segment_max(data.numpy(), lens.numpy()))
    out.sum().backward()
    xc = data.clone().requires_grad_(True)
    torch.segment_reduce(xc, 'max', lengths=lens, unsafe=True, initial=float(data.min())).sum().backward()
    torch.testing.assert_close(x.grad.cpu(), xc.grad)
    assert _scratch_is_clean()


def test_the_scan_counts_the_empty_sequences_and_the_reduce_believes_it():
    """rua_exclusive_scan_i64 leaves the number of inputs <= 0 in total[1] (one tile, many tiles); a CAT layout built from
    device-only lengths carries that word (rua_layout::bsz) and max / min / logsumexp skip the tracking of the global
    extreme when it reads 0.  The word follows the lengths: writing a zero into them in place (a version bump) brings a
    fresh scan, a fresh count and the tracking back."""
    from torchrua_amd import _meta as M
    g = torch.Generator().manual_seed(21)
    for n in (5, 2048, 2049, 70_001):
        x = torch.randint(-2, 6, (n,), generator=g)
        off, total = M.exclusive_scan(x.to(DEV), want_total=True)
        assert total.cpu().tolist() == [int(x.sum()), int((x <= 0).sum())], n
        assert torch.equal(off.cpu(), torch.cumsum(x, 0) - x)
    lens = torch.randint(1, 30, (3000,), generator=g)
    data = torch.randn(int(lens.sum()), 32, generator=g)
    data[:5] -= 50.0                                  # the global minimum sits in rows every variant below still reads
    dd, ld = data.to(DEV), lens.to(DEV)
    out = ta.segment_logsumexp(dd, ld)
    assert int(M.dev_n_empty(ld)) == 0                # nothing empty: this call did not track
    np.testing.assert_allclose(out.cpu().numpy(), orc.segment_logsumexp(data.numpy(), lens.numpy()), rtol=2e-5, atol=2e-5)
    ld[1234] = 0                                      # in place: lengths now sum to less than the payload holds
    lens2 = lens.clone()
    lens2[1234] = 0
    n2 = int(lens2.sum())
    for name in ('max', 'logsumexp'):
        got = getattr(ta, f'segment_{name}')(dd, ld).cpu().numpy()
        assert int(M.dev_n_empty(ld)) == 1
        ref = getattr(orc, f'segment_{name}')(data[:n2].numpy(), lens2.numpy())
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5, err_msg=name)
        assert np.all(got[1234] == data.min().item())
    assert _scratch_is_clean()
