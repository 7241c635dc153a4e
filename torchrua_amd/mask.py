"""mask / bmask / fmask (the API of torchrua.mask, reference mask.py:6-38).  The reference fills a [b, t] grid
and scatters `one` through an N-element index; here ONE kernel writes every cell (`t < len[b] ? one : zero`,
rua_mask) — an attention-mask producer usually sits right behind a pad, on grids of B x T cells."""
import torch

from torchrua_amd.core import _mask_grid
from torchrua_amd.layout import C, L, P, R, T, Z, lens_of

__all__ = []  # methods are attached to the layout classes


def _mask(self: Z, zero, one, dtype: torch.dtype = None) -> T:
    lens = lens_of(self)        # (one row per sequence of the batch, zero-length ones included)
    return _mask_grid(lens, lens.numel(), self.size()[1], zero, one, self.data.dtype if dtype is None else dtype)


# the two fixed flavours: which (zero, one, dtype) a container asks `mask` for
_FLAVOURS = {
    'bmask': lambda z: (False, True, torch.bool),                                    # True on tokens
    'fmask': lambda z: (torch.finfo(z.data.dtype).min, 0, z.data.dtype),             # additive: 0 on tokens, -max off
}


def _flavoured(name: str):
    def method(self: Z) -> T:
        zero, one, dtype = _FLAVOURS[name](self)
        return self.mask(zero=zero, one=one, dtype=dtype)
    method.__name__ = name
    return method


for _cls in (C, L, P, R):
    _cls.mask = _mask
    for _name in _FLAVOURS:
        setattr(_cls, _name, _flavoured(_name))
