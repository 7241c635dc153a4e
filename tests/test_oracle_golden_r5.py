"""The oracle against the round-5 fixtures (tests/golden/r5.npz, `oracle/gen_golden.py --round5`, from the reference):
integer scatter_logsumexp (reduce.py:26-31 answers in float32).  The `layout.r5.*` cases of the same file — 1-D payloads
of 1 / 2 / 4 / 8-byte rows, rows of 8 / 4 / 2 (mod 16) bytes — are walked by tests/test_oracle_golden.py."""
import numpy as np
import pytest

from helpers import cases, golden, orc


@pytest.mark.parametrize('case', cases('scatter_lse_int.'))
def test_integer_scatter_logsumexp(case):
    g = golden()[case]
    for inc in (0, 1):
        want = g[f'scatter_logsumexp.{inc}']
        got = orc.scatter_logsumexp(g['tensor'], g['index'], g['source'], include_self=bool(inc))
        assert got.dtype == want.dtype == np.float32
        np.testing.assert_allclose(got, want, rtol=3e-6, atol=3e-6, err_msg=f'{case} include_self={inc}')
        assert np.array_equal(np.isinf(got), np.isinf(want))


def test_the_fixtures_cover_what_they_claim():
    g = golden()
    assert {g[c]['tensor'].dtype for c in cases('scatter_lse_int.')} == {np.dtype(k) for k in (np.int64, np.int32, np.int16, np.int8)}
    rows = {c: g[c]['data'].dtype.itemsize * int(np.prod(g[c]['data'].shape[1:], dtype=np.int64)) for c in cases('layout.r5.')}
    assert {1, 2, 4, 8} <= set(rows.values())                                  # sub-16-byte rows, 1-D and 2-D
    assert {1000, 2000, 500, 18} <= set(rows.values())                         # rows of 8 / 4 / 2 (mod 16) bytes
    assert g['layout.r5.vec.bool']['data'].dtype == np.bool_
    i8 = g['scatter_lse_int.i8']                                               # differences that wrap in int8
    assert int(i8['source'].max()) - int(i8['source'].min()) > 127
