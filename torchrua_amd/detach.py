"""split / tolist (the API of torchrua.detach, reference detach.py:9-51): hand the sequences back as a Python
list.  Inherently host-side and synchronous; the only device work is the P -> C move.

C/P: one torch.split by the (host-mirrored, if available) lengths.  L/R: a sequence is a slice of its own padded
row, so the list is B views `data[b, lo:hi]` — no flattening of the padded storage, no sections for the padding."""
from typing import List

import torch

from torchrua_amd import _meta as M
from torchrua_amd.layout import C, L, P, R, T, Z

__all__ = []  # methods are attached to the layout classes


def _split_dense(self) -> List[T]:
    dense = self.cat()
    return list(torch.split(dense.data, M.host_lens(dense.token_sizes).tolist()))


def _split_left(self: L) -> List[T]:
    return [self.data[b, :n] for b, n in enumerate(M.host_lens(self.token_sizes).tolist())]


def _split_right(self: R) -> List[T]:
    n_steps = self.size()[1]                     # right alignment is against the longest sequence
    return [self.data[b, n_steps - n:n_steps] for b, n in enumerate(M.host_lens(self.token_sizes).tolist())]


def _tolist(self: Z):
    """(the reference's P.tolist fails — PackedSequence has no detach(); this one works for all four)"""
    return [piece.detach().cpu().tolist() for piece in self.split()]


C.split = P.split = _split_dense
L.split = _split_left
R.split = _split_right
for _cls in (C, L, P, R):
    _cls.tolist = _tolist
