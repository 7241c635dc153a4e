"""mask / bmask / fmask — mirror of torchrua.mask (reference mask.py:6-38): one kernel writes the whole
[b, t] grid (`t < len[b] ? one : zero`) instead of new_full + an N-element index_put."""
import torch

from torchrua_amd.core import _mask_grid
from torchrua_amd.layout import C, L, P, R, T, Z, lens_of


__all__ = []  # methods are attached to the layout classes


def _mask(self: Z, zero, one, dtype: torch.dtype = None) -> T:
    """mask.py:6-14."""
    b, t = self.size()[:2]
    return _mask_grid(lens_of(self), b, t, zero, one, self.data.dtype if dtype is None else dtype)


def _bmask(self: Z) -> T:
    """mask.py:22-23."""
    return self.mask(zero=False, one=True, dtype=torch.bool)


def _fmask(self: Z) -> T:
    """mask.py:31-32."""
    return self.mask(zero=torch.finfo(self.data.dtype).min, one=0, dtype=self.data.dtype)


for _cls in (C, L, P, R):
    _cls.mask = _mask
    _cls.bmask = _bmask
    _cls.fmask = _fmask
