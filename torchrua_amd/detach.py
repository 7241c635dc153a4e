"""split / tolist — mirror of torchrua.detach (reference detach.py:9-51): host-side list
materialisation (inherently synchronous)."""
from typing import List

import torch

from torchrua_amd import _meta as M
from torchrua_amd.layout import C, L, P, R, T, Z


__all__ = []  # methods are attached to the layout classes


def _cat_pack_split(self) -> List[T]:
    """detach.py:9-13."""
    data, token_sizes = self.cat()
    return torch.split(data, M.host_lens(token_sizes).tolist(), dim=0)


def _padded_split(right: bool):
    def split(self) -> List[T]:
        """detach.py:20-27 / 33-40."""
        t = self.size()[1]
        lens = M.host_lens(self.token_sizes)
        pair = [t - lens, lens] if right else [lens, t - lens]
        sections = torch.stack(pair, dim=-1).view(-1).tolist()
        return torch.split(self.data.flatten(start_dim=0, end_dim=1), sections, dim=0)[(1 if right else 0)::2]
    return split


def _tolist(self: Z):
    """detach.py:44-45 (the reference's P.tolist is broken: PackedSequence has no detach(); ours works)."""
    return [tensor.detach().cpu().tolist() for tensor in self.split()]


C.split = _cat_pack_split
P.split = _cat_pack_split
L.split = _padded_split(False)
R.split = _padded_split(True)
for _cls in (C, L, P, R):
    _cls.tolist = _tolist
