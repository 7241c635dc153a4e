"""Pins the oracle's restatements of compose, split and Z- / tensor-keyed indexing against the reference's own outputs
(tests/golden/r3.npz, oracle/gen_golden.py --round3).  CPU only.  (The gradient fixtures of r3.npz have no oracle
counterpart: the GPU tests compare the kernels' backward with them directly.)"""
import numpy as np
import pytest
import torch

from helpers import assert_seq_equal, cases, golden, orc

KINDS = 'CLPR'


def _sort_desc(lens):
    """The reference's host call (core/view.py:48)."""
    return torch.sort(torch.from_numpy(np.ascontiguousarray(lens)), descending=True)[1].numpy()


def _fill_for(f):
    if f['data'].dtype == np.uint16:
        return np.array([0xBFC0], dtype=np.uint16)        # bf16 bits of -1.5
    return -7 if f['data'].dtype.kind == 'i' else -1.5


def _as_kind(c, k, fill, srt):
    return orc.to_kind(c, k, fill, srt)


@pytest.mark.parametrize('case', cases('zkey.'))
def test_z_and_tensor_keys(case):
    f = golden()[case]
    fill = _fill_for(f)
    c = orc.C(f['data'], f['lens'])
    seqs = {k: _as_kind(c, k, fill, f['sorted_indices']) for k in KINDS}
    klens = f['key.lens']
    ksrt = _sort_desc(klens)
    for k, z in seqs.items():
        krows, uniq, value = f[f'key.{k}.rows'], f[f'key.{k}.uniq'], f[f'key.{k}.value']
        for kz in KINDS:
            key = _as_kind(orc.C(krows, klens), kz, 0, ksrt)
            assert_seq_equal(orc.getitem(z, key), f, f'getitem_z.{k}.{kz}', kz)
            ukey = _as_kind(orc.C(uniq, klens), kz, 0, ksrt)
            z2 = z.with_data(z.data.copy())
            if kz in 'CP':
                orc.setitem(z2, ukey, _as_kind(orc.C(value, klens), kz, 0, ksrt).data)
            else:
                orc.setitem(z2, ukey, np.asarray(3).astype(z.data.dtype) if z.data.dtype != np.uint16
                            else np.array(0x4040, dtype=np.uint16))          # bf16 bits of 3.0
            np.testing.assert_array_equal(z2.data, f[f'setitem_z.{k}.{kz}'], err_msg=f'setitem_z.{k}.{kz}')
            if k == 'C':
                assert_seq_equal(orc.getitem(orc.C(f['data'], f['lens']), key), f, f'tensor_getitem.{kz}', kz)
                np.testing.assert_array_equal(z2.data, f[f'tensor_setitem.{kz}'])
        np.testing.assert_array_equal(orc.getitem(z, krows), f[f'getitem_t.{k}.1d'])
        two = krows[:krows.size // 2 * 2].reshape(2, -1)
        np.testing.assert_array_equal(orc.getitem(z, two), f[f'getitem_t.{k}.2d'])
        z3 = z.with_data(z.data.copy())
        orc.setitem(z3, uniq, value)
        np.testing.assert_array_equal(z3.data, f[f'setitem_t.{k}'])


@pytest.mark.parametrize('case', cases('split.'))
def test_split(case):
    f = golden()[case]
    lens = f['lens']
    arrays = np.split(f['data'], np.cumsum(lens)[:-1])
    srt = _sort_desc(lens)
    for k in KINDS:
        z = orc.new(k, arrays, 0, srt)
        parts = orc.split(z)
        np.testing.assert_array_equal(np.asarray([p.shape[0] for p in parts]), f[f'split.{k}.sizes'])
        np.testing.assert_array_equal(np.concatenate(parts), f[f'split.{k}.cat'])
    for k in 'LR':                       # storage wider than the longest sequence: the reference's split raises
        wide = orc.Seq(k, f[f'wide.{k}.data'], token_sizes=lens)
        with pytest.raises(RuntimeError):
            orc.split(wide)
        assert_seq_equal(orc.to_cat(wide), f, f'wide.{k}.cat', 'C')


@pytest.mark.parametrize('case', cases('compose.'))
def test_compose(case):
    f = golden()[case]
    seqs = []
    for i in range(int(f['n'])):
        kind = bytes(f[f'in{i}.kind']).decode()
        lens = f[f'in{i}.lens']
        arrays = np.split(f[f'in{i}.data'], np.cumsum(lens)[:-1])
        seqs.append(orc.new(kind, arrays, 0, _sort_desc(lens)))
    assert_seq_equal(orc.compose(seqs, _sort_desc), f, 'out', 'P')


@pytest.mark.parametrize('case', cases('view.'))
def test_views(case):
    """core/view.py:21-77: cat_view / left_view / pack_view / right_view of every layout."""
    f = golden()[case]
    fill = _fill_for(f)
    c = orc.C(f['data'], f['lens'])
    srt = f['sorted_indices']
    for k in KINDS:
        z = _as_kind(c, k, fill, srt)
        assert_seq_equal(orc.cat_view(z), f, f'view.{k}.C', 'C')
        assert_seq_equal(orc._padded_view(z, 'L', fill), f, f'view.{k}.L', 'L')
        assert_seq_equal(orc._padded_view(z, 'R', fill), f, f'view.{k}.R', 'R')
        assert_seq_equal(orc.pack_view(z, srt), f, f'view.{k}.P', 'P')
        assert_seq_equal(orc._padded_view(z, 'L', 7, np.int64), f, f'view.{k}.L.long', 'L')


@pytest.mark.parametrize('case', cases('scatterdim.'))
def test_scatter_along_another_dim(case):
    """scatter_*(..., dim != 0): the oracle reduces along rows, so `dim` goes to the front and back (what
    torch.index_reduce's `dim` means)."""
    f = golden()[case]
    idx = f['index']
    for tag in ('last', 'neg', 'mid'):
        ten, src, dim = f[f'{tag}.tensor'], f[f'{tag}.source'], int(f[f'{tag}.dim'])
        t0, s0 = np.ascontiguousarray(np.moveaxis(ten, dim, 0)), np.ascontiguousarray(np.moveaxis(src, dim, 0))
        for name in ('max', 'min', 'sum', 'mean', 'prod', 'logsumexp'):
            for inc in (0, 1):
                got = getattr(orc, f'scatter_{name}')(t0.reshape(t0.shape[0], -1), idx, s0.reshape(s0.shape[0], -1),
                                                      include_self=bool(inc)).reshape(t0.shape)
                got = np.moveaxis(got, 0, dim)
                if f'{tag}.scatter_{name}.{inc}' not in f:
                    continue          # the reference's scatter_logsumexp raises for dim != 0 (META.json)
                exp = f[f'{tag}.scatter_{name}.{inc}']
                if name in ('max', 'min'):
                    np.testing.assert_array_equal(got, exp, err_msg=f'{tag}.{name}.{inc}')
                else:
                    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-6, err_msg=f'{tag}.{name}.{inc}')


@pytest.mark.parametrize('case', cases('maskall.'))
def test_masks_of_every_layout(case):
    f = golden()[case]
    fill = _fill_for(f)
    bf = f['data'].dtype == np.uint16
    c = orc.C(f['data'], f['lens'])
    srt = _sort_desc(f['lens'])
    for k in KINDS:
        z = _as_kind(c, k, fill, srt)
        np.testing.assert_array_equal(orc.mask(z, False, True, np.bool_), f[f'bmask.{k}'])
        np.testing.assert_array_equal(orc.mask(z, -3, 9, np.int32), f[f'mask.{k}.i32'])
        np.testing.assert_array_equal(orc.mask(z, 7, 1, np.uint8), f[f'mask.{k}.u8'])
        if not bf:
            np.testing.assert_array_equal(orc.mask(z, np.finfo(np.float32).min, 0, np.float32), f[f'fmask.{k}'])
            np.testing.assert_array_equal(orc.mask(z, 0.5, -2.0, np.float32), f[f'mask.{k}.own'])
