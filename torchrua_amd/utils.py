"""Index primitives — mirror of torchrua.utils (reference utils.py:7-26), each one kernel."""
from typing import Any, List, Tuple

import torch
from torch import Tensor

from torchrua_amd import _lib as K
from torchrua_amd import _meta as M

__all__ = ['major_sizes_to_ptr', 'get_offsets', 'invert_permutation']   # (what the reference's other modules import)


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
    M._memo_put(sizes, 'n_empty_dev', total[1:2])
    n = M._memo_get(sizes, 'sum')
    if n is None:
        n = int(total[0].item())  # the expansion's length is data-dependent: one sync (the reference's
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


# ---- utils.py:33-51: shape helpers the reference defines and never calls (dead code there).  Kept for
# the drop-in namespace only; they are shape arithmetic around torch.gather, not part of the hot path.
def with_shape(shape, dim: int, value: int) -> List[int]:
    """utils.py:33-36: `shape` with entry `dim` replaced."""
    out = [int(d) for d in shape]
    out[dim] = value
    return out


def broadcast_shapes(*sizes, dim: int) -> List[List[int]]:
    """utils.py:39-41: broadcast every dimension but `dim`, which each shape keeps."""
    common = torch.broadcast_shapes(*(with_shape(size, dim, 1) for size in sizes))
    return [with_shape(common, dim, size[dim]) for size in sizes]


def broadcast_tensors(*tensors: Tensor, dim: int) -> List[Tensor]:
    """utils.py:44-46."""
    return [t.expand(shape) for t, shape in zip(tensors, broadcast_shapes(*(t.size() for t in tensors), dim=dim))]


def gather(tensor: Tensor, index: Tensor, dim: int) -> Tensor:
    """utils.py:49-51."""
    tensor, index = broadcast_tensors(tensor, index, dim=dim)
    return tensor.gather(dim=dim, index=index)
