"""compose(sequences): ONE PackedSequence over the sequences of several containers (the API of
torchrua.compose, reference compose.py:9-33; SURVEY.md §8f rank 4 — a caller next to the hot path: it batches
several containers into one LSTM input).

Batch order of the result: the containers interleaved — sequence 0 of every container, then sequence 1 of
every container that has one, ... with the containers ordered as packing them would order them (more
sequences first).  That order depends on nothing but the container sizes, which the host knows, so it is
computed there (numpy) instead of by packing an index container on the device; the payload then moves ONCE, one
scatter-mode launch of the row mover per container, from each container's own storage into the row order of an
ordinary pack() of all the sequences (the reference concatenates every storage first: compose.py:33)."""
from typing import List

import numpy as np
import torch

from torchrua_amd import _lib as K
from torchrua_amd import _meta as M
from torchrua_amd import _ops as O
from torchrua_amd.core import _gather_flat, _scatter_flat
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


class _ComposeRows(torch.autograd.Function):
    """out[j] = row of the container that composed row j comes from — ONE scatter-mode launch of the row mover per
    container, straight from that container's own storage (no concatenated copy of the storages: VERDICT r2 weak #7).
    `inv` maps every storage row (containers back to back) to its composed row, or to n_out for a padding row, which the
    mover then skips.  The adjoint gathers the cotangent's rows back per container (padding rows: zero)."""

    @staticmethod
    def forward(ctx, inv: torch.Tensor, n_out: int, *storages: torch.Tensor):
        first = storages[0]
        out = torch.empty((n_out,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
        base, parts = 0, []
        for st in storages:
            rows = int(st.size(0))
            part = inv[base:base + rows]
            parts.append(part)
            if rows:
                plan = O.MovePlan(M.lay_list(None, part), M.lay_flat(n_out), out.shape, flags=K.MOVE_SCATTER, name='compose')
                O.launch_move(plan, st.detach(), out=out)
            base += rows
        ctx.parts, ctx.n_out = parts, n_out
        ctx.shapes = [tuple(st.shape) for st in storages]
        return out

    @staticmethod
    def backward(ctx, grad: torch.Tensor):
        # (a gather per container; [r4] through the differentiable gather, so compose is twice differentiable like the
        # reference's, which is casts and a cat)
        grad = grad.contiguous()
        outs = []
        for part, shape in zip(ctx.parts, ctx.shapes):
            if shape[0] == 0:
                outs.append(grad.new_zeros(shape))
                continue
            outs.append(O.gather_rows(grad, part).reshape(shape))
        return (None, None) + tuple(outs)


def compose(sequences: List[Z]) -> P:
    storages, rows, lens, base = [], [], [], 0
    for z in sequences:                                                # token -> row of the storages, back to back
        flat = z.idx().cat()
        storages.append(z.raw())
        rows.append(flat.data + base)
        lens.append(flat.token_sizes)
        base += int(storages[-1].size(0))
    dev = K.require_device(*storages)
    hidden, dtype = tuple(storages[0].shape[1:]), storages[0].dtype
    if any(tuple(st.shape[1:]) != hidden or st.dtype != dtype for st in storages):
        raise K.RuaError('compose: the containers must agree in dtype and trailing shape (torch.cat would refuse them too)')
    order = M.to_device_async(torch.from_numpy(_interleaved([int(n.numel()) for n in lens])), dev)

    packed = C(data=torch.cat(rows), token_sizes=torch.cat(lens)).pack()      # packs the row NUMBERS (8 bytes a token)
    unsorted = _gather_flat(packed.unsorted_indices, order)                   # re-index the batch by the composed order
    n_out = int(packed.data.numel())
    inv = torch.full((base,), n_out, dtype=torch.long, device=dev)            # storage row -> composed row (n_out: none)
    _scatter_flat(inv, packed.data, torch.arange(n_out, dtype=torch.long, device=dev))
    payload = _ComposeRows.apply(inv, n_out, *[st.contiguous() for st in storages])
    return P(data=payload, batch_sizes=packed.batch_sizes, sorted_indices=invert_permutation(unsorted),
             unsorted_indices=unsorted)
