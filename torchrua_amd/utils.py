"""Index primitives — mirror of torchrua.utils (reference utils.py:7-26), each one kernel."""
from typing import Tuple

import torch
from torch import Tensor

from torchrua_amd import _lib as K
from torchrua_amd import _meta as M

__all__ = ['major_sizes_to_ptr', 'get_offsets', 'invert_permutation']


def get_offsets(sizes: Tensor) -> Tensor:
    """utils.py:16-19: exclusive prefix sum (K1, wavefront scan)."""
    return M.exclusive_scan(M._as_lens(sizes))


def major_sizes_to_ptr(sizes: Tensor) -> Tuple[Tensor, Tensor]:
    """utils.py:7-13: (position inside its run, run id) for every element of the expansion (K2)."""
    dev = K.require_device(sizes)
    lib = K.load()
    sizes = M._as_lens(sizes)
    off, total = M.exclusive_scan(sizes, want_total=True)
    M._memo_put(sizes, 'off', off)
    n = M._memo_get(sizes, 'sum')
    if n is None:
        n = int(total.item())     # the expansion's length is data-dependent: one sync (the reference's
        M._memo_put(sizes, 'sum', n)  # repeat_interleave syncs here too)
    lay = M.lay_cat(sizes, sizes.numel(), n)
    major = torch.empty(n, dtype=torch.long, device=dev)
    minor = torch.empty(n, dtype=torch.long, device=dev)
    K.check(lib.rua_enum_rows(lay.ref(), n, K.ptr(minor), K.ptr(major), None, K.stream_ptr(dev)), 'rua_enum_rows')
    return major, minor


def invert_permutation(tensor: Tensor) -> Tensor:
    """utils.py:22-26: inv[p[i]] = i (K3 with no length work)."""
    dev = K.require_device(tensor)
    lib = K.load()
    p = M._as_lens(tensor)
    inv = torch.empty_like(p)
    K.check(lib.rua_pack_meta(None, K.ptr(p), p.numel(), 0, K.ptr(inv), None, K.stream_ptr(dev)), 'rua_pack_meta')
    return inv
