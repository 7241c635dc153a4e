"""HIP path vs tests/golden/r5.npz (the reference's own outputs, oracle/gen_golden.py --round5): integer
scatter_logsumexp.  (The `layout.r5.*` cases — sub-16-byte and 8-mod-16-byte rows through every layout / select
function — run under tests/test_gpu_golden.py, which walks every `layout.` case.)"""
import numpy as np
import pytest
import torch

import torchrua_amd as ta
from gpu_util import DEV
from helpers import cases, golden, to_np, to_torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', cases('scatter_lse_int.'))
def test_integer_scatter_logsumexp(case):
    """reduce.py:26-31 on integer tensors: float32 out, 1e-5 relative of the reference (tolerance of north_star)."""
    f = golden()[case]
    ten, idx, src = to_torch(f['tensor'], DEV), to_torch(f['index'], DEV), to_torch(f['source'], DEV)
    for inc in (0, 1):
        got = ta.scatter_logsumexp(ten, idx, src, include_self=bool(inc))
        want = f[f'scatter_logsumexp.{inc}']
        assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
        g = to_np(got)
        assert np.array_equal(np.isinf(g), np.isinf(want)), f'{case} {inc}'
        fin = np.isfinite(want)
        np.testing.assert_allclose(g[fin], want[fin], rtol=1e-5, atol=1e-5, err_msg=f'{case} include_self={inc}')
    assert to_np(ten).tobytes() == f['tensor'].tobytes()


def test_uint8_scatter_logsumexp_is_refused():
    ten = torch.zeros(4, 2, dtype=torch.uint8, device=DEV)
    with pytest.raises(ta.RuaError):
        ta.scatter_logsumexp(ten, torch.zeros(3, dtype=torch.long, device=DEV), torch.ones(3, 2, dtype=torch.uint8, device=DEV))
