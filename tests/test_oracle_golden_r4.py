"""The oracle against the round-4 fixtures (tests/golden/r4.npz, generated from the reference by
`oracle/gen_golden.py --round4`): scatter_* on INTEGER tensors (reduce.py:6-23), bit-exact."""
import numpy as np
import pytest

from helpers import cases, golden, orc

OPS = ('max', 'min', 'sum', 'mean', 'prod')


@pytest.mark.parametrize('case', cases('scatter_int.'))
def test_integer_scatter(case):
    g = golden()[case]
    tensor, index, source = g['tensor'], g['index'], g['source']
    for name in OPS:
        for inc in (0, 1):
            want = g[f'scatter_{name}.{inc}']
            got = getattr(orc, f'scatter_{name}')(tensor, index, source, include_self=bool(inc))
            assert got.dtype == want.dtype == tensor.dtype
            np.testing.assert_array_equal(got, want, err_msg=f'{case} scatter_{name} include_self={inc}')
    if 'last.tensor' in g:            # dim = -1 of an [H, S] target: the same reduce over the transposed operands
        tt, ss = g['last.tensor'], g['last.source']
        for name in OPS:
            for inc in (0, 1):
                got = getattr(orc, f'scatter_{name}')(np.ascontiguousarray(tt.T), index, np.ascontiguousarray(ss.T),
                                                      include_self=bool(inc)).T
                np.testing.assert_array_equal(got, g[f'last.scatter_{name}.{inc}'], err_msg=f'{case} last {name} {inc}')


def test_the_fixtures_cover_what_they_claim():
    g = golden()
    assert len(cases('scatter_int.')) >= 15
    kinds = {g[c]['tensor'].dtype for c in cases('scatter_int.')}
    assert kinds == {np.dtype(k) for k in (np.int64, np.int32, np.int16, np.int8, np.uint8)}
    wraps = g['scatter_int.i8.count_wraps']
    assert np.bincount(wraps['index']).max() > 255          # the int8 count of `mean` wraps past 127 and past 255
