"""GPU-side helpers for the parity tests: golden/oracle containers <-> torchrua_amd containers."""
import numpy as np
import torch

import torchrua_amd as ta
from helpers import orc, to_np, to_torch

DEV = torch.device('cuda:0')
KINDS = {'C': ta.C, 'L': ta.L, 'P': ta.P, 'R': ta.R}


def kind_of(z) -> str:
    for k, cls in KINDS.items():
        if isinstance(z, cls):
            return k
    raise TypeError(type(z))


def dev_seq(s: orc.Seq, bf16=False):
    """oracle Seq (numpy) -> torchrua_amd container on the GPU."""
    data = to_torch(s.data, DEV, bf16=bf16)
    if s.kind == 'P':
        return ta.P(data, torch.from_numpy(s.batch_sizes.copy()), to_torch(s.sorted_indices, DEV),
                    to_torch(s.unsorted_indices, DEV))
    return KINDS[s.kind](data, to_torch(s.token_sizes, DEV))


def host_seq(z) -> orc.Seq:
    """torchrua_amd container -> oracle Seq (numpy; bf16 as uint16 bits)."""
    k = kind_of(z)
    if k == 'P':
        return orc.P(to_np(z.data), to_np(z.batch_sizes), to_np(z.sorted_indices), to_np(z.unsorted_indices))
    return orc.Seq(k, to_np(z.data), token_sizes=to_np(z.token_sizes))


def assert_same_seq(actual, expected: orc.Seq, what='', exact=True, rtol=0.0, atol=0.0, valid_only=False):
    a = host_seq(actual)
    assert a.kind == expected.kind, f'{what}: kind {a.kind} vs {expected.kind}'
    for f in ('token_sizes', 'batch_sizes', 'sorted_indices', 'unsorted_indices'):
        e = getattr(expected, f)
        if e is not None:
            np.testing.assert_array_equal(getattr(a, f), e, err_msg=f'{what}.{f}')
    ad, ed = a.data, np.asarray(expected.data)
    assert ad.shape == ed.shape, f'{what}: shape {ad.shape} vs {ed.shape}'
    assert ad.dtype == ed.dtype, f'{what}: dtype {ad.dtype} vs {ed.dtype}'
    if valid_only and a.kind in 'LR':
        m = orc.get_mask(expected).astype(bool)
        if a.kind == 'R':
            m = m[:, ::-1]
        ad, ed = ad[m], ed[m]
    if exact:
        assert np.array_equal(ad, ed, equal_nan=ad.dtype.kind == 'f'), f'{what}: payload not bit-exact'
    else:
        np.testing.assert_allclose(ad, ed, rtol=rtol, atol=atol, equal_nan=True, err_msg=what)


def host_sort(lens) -> np.ndarray:
    """The reference's host call for sorted_indices (core/view.py:48), on THIS machine."""
    t = torch.as_tensor(np.asarray(lens), dtype=torch.long)
    return torch.sort(t, descending=True)[1].numpy()
