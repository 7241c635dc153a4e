"""compose(sequences) -> one PackedSequence over the sequences of several containers — mirror of
torchrua.compose (reference compose.py:9-33; SURVEY.md §8f rank 4, a caller next to the hot path).
Index bookkeeping through the same kernels; one flat row gather moves the payload."""
from typing import List

import torch

from torchrua_amd.core import _gather_flat, _new_cat
from torchrua_amd.layout import C, P, Z
from torchrua_amd.utils import invert_permutation

__all__ = ['compose']


def compose(sequences: List[Z]) -> P:
    offset, data, indices, token_sizes = 0, [], [], []
    for sequence in sequences:
        raw = sequence.raw()
        data.append(raw)
        idx, sizes = sequence.idx().cat()
        indices.append(idx + offset)
        token_sizes.append(sizes)
        offset += raw.size()[0]

    groups = _new_cat(token_sizes)                       # data = all lengths, token_sizes = #seqs per container
    order = groups.idx().pack().data                     # container-interleaved order of the sequences
    packed = C(data=torch.cat(indices, dim=0), token_sizes=groups.data).pack()
    unsorted_indices = _gather_flat(packed.unsorted_indices, order)
    packed = packed._replace(sorted_indices=invert_permutation(unsorted_indices), unsorted_indices=unsorted_indices)
    return packed._replace(data=_gather_flat(torch.cat(data, dim=0), packed.data))
