"""compose(sequences): ONE PackedSequence over the sequences of several containers (the API of
torchrua.compose, reference compose.py:9-33; SURVEY.md §8f rank 4 — a caller next to the hot path: it batches
several containers into one LSTM input).

Batch order of the result: the containers interleaved — sequence 0 of every container, then sequence 1 of
every container that has one, ... with the containers ordered as packing them would order them (more
sequences first).  That order depends on nothing but the container sizes, which the host knows, so it is
computed there (numpy) instead of by packing an index container on the device; the payload is then ONE
flat row gather (rua_move_rows against a LIST layout) in the row order of an ordinary pack() of all the
sequences."""
from typing import List

import numpy as np
import torch

from torchrua_amd import _meta as M
from torchrua_amd.core import _gather_flat
from torchrua_amd.layout import C, P, Z
from torchrua_amd.utils import invert_permutation

__all__ = ['compose']


def _interleaved(counts: List[int]) -> np.ndarray:
    """Position j of the composed batch -> index of that sequence in container-major numbering."""
    counts_t = torch.tensor(counts, dtype=torch.long)
    rank = M.host_sort_desc(counts_t).numpy()  # the order pack() gives containers: its own host sort, same tie order
    first = np.concatenate(([0], np.cumsum(counts)[:-1]))
    step = np.arange(max(counts))[:, None]                             # [steps, 1]
    grid = first[rank][None, :] + step                                 # sequence `step` of container rank[k]
    return grid[step < np.asarray(counts)[rank][None, :]]              # row-major: step-major, containers in rank order


def compose(sequences: List[Z]) -> P:
    storages, rows, lens, base = [], [], [], 0
    for z in sequences:                                                # token -> row of the concatenated storages
        flat = z.idx().cat()
        storages.append(z.raw())
        rows.append(flat.data + base)
        lens.append(flat.token_sizes)
        base += int(storages[-1].size(0))
    dev = storages[0].device
    order = M.to_device_async(torch.from_numpy(_interleaved([int(n.numel()) for n in lens])), dev)

    packed = C(data=torch.cat(rows), token_sizes=torch.cat(lens)).pack()      # packs the row numbers themselves
    unsorted = _gather_flat(packed.unsorted_indices, order)                   # re-index the batch by the composed order
    payload = _gather_flat(torch.cat(storages), packed.data)
    return P(data=payload, batch_sizes=packed.batch_sizes, sorted_indices=invert_permutation(unsorted),
             unsorted_indices=unsorted)
