"""head / last / roll / rev / trunc — mirror of torchrua.select (reference select/head.py, last.py,
roll.py, rev.py, trunc.py).  Each op that moves payload is ONE launch of the row mover with a
closed-form per-sequence token map (include/rua.h: enum rua_tmap); where the reference returns a
view (P.head, L.head, L/R.trunc) so do we.
"""
from typing import Tuple

import torch
from torch import Tensor

from torchrua_amd import _lib as K
from torchrua_amd import _meta as M
from torchrua_amd import _ops as O
from torchrua_amd.core import _hidden, pack_reference_order
from torchrua_amd.layout import C, L, P, R, Z, describe, lens_of


__all__ = []  # methods are attached to the layout classes


# ------------------------------------------------------------------ head (select/head.py)
def _cat_head(self: C, n: int) -> C:
    """select/head.py:6-16 (.tolist() + 2B split views + cat in the reference)."""
    B = self.token_sizes.size(0)
    dst = M.lay_cat(None, B, B * n, len_add=n)
    data = O.move(self.data, O.MovePlan(dst, describe(self), (B * n,) + _hidden(self), name='head'))
    return C(data=data, token_sizes=torch.full_like(self.token_sizes, fill_value=n))


def _pack_head(self: P, n: int) -> P:
    """select/head.py:22-30: zero-copy slice (valid for n <= min length, as in the reference)."""
    data, batch_sizes, sorted_indices, unsorted_indices = self
    return P(data=data[:M.pack_B(self) * n], batch_sizes=batch_sizes[:n], sorted_indices=sorted_indices,
             unsorted_indices=unsorted_indices)


def _left_head(self: L, n: int) -> L:
    """select/head.py:36-42: a view."""
    return L(data=self.data[:, :n], token_sizes=torch.full_like(self.token_sizes, n))


def _right_head(self: R, n: int) -> R:
    """select/head.py:48-64: rows (t_phys - len[b]) .. +n of every sequence (the reference aligns on the
    PHYSICAL t here: `b, t, *_ = data.size()`)."""
    B, t_phys = self.data.shape[:2]
    src = M.lay_padded(K.RIGHT, self.token_sizes, B, t_phys, t_phys)
    dst = M.lay_padded(K.RIGHT, None, B, n, n, len_add=n)
    data = O.move(self.data, O.MovePlan(dst, src, (B, n) + _hidden(self), name='head'))
    return R(data=data, token_sizes=torch.full_like(self.token_sizes, fill_value=n))


C.head = _cat_head
P.head = _pack_head
L.head = _left_head
R.head = _right_head


# ------------------------------------------------------------------ last (select/last.py:7-19)
def _last(self: Z) -> Tensor:
    """One row per sequence: token len[b]-1 = REV_S of token 0 of a length-1 destination."""
    lens = lens_of(self)
    B = lens.numel()
    dst = M.lay_padded(K.LEFT, None, B, 1, 1, len_add=1)
    return O.move(self.data, O.MovePlan(dst, describe(self), (B,) + _hidden(self), tmap=K.T_REV_S, name='last'))


for _cls in (C, L, P, R):
    _cls.last = _last


# ------------------------------------------------------------------ roll / rev
def _same_layout_move(self: Z, tmap: int, arg: int, name: str, pad_row: int = -1, logical_stride: bool = False):
    """Permute tokens inside every sequence, layout unchanged (metadata tensors are shared)."""
    if isinstance(self, (L, R)):
        # the reference builds these through .left()/.right() of an intermediate: the result has the
        # LOGICAL t = token_sizes.max() rows per sequence, padding = 0
        b, t = self.size()[:2]
        kind = K.LEFT if isinstance(self, L) else K.RIGHT
        dst = M.lay_padded(kind, self.token_sizes, b, t, t)
        # roll goes through self.idx(), which strides the flat storage by the logical t
        # (layout/left.py:73-77); rev reads data[b, ...] with the physical stride (select/rev.py:25-41)
        stride = t if logical_stride else int(self.data.size(1))
        src = M.lay_padded(kind, self.token_sizes, b, stride, t)
        plan = O.MovePlan(dst, src, (b, t) + _hidden(self), tmap, arg, fill=0, pad_row=pad_row, name=name)
    else:
        lay = describe(self)
        order = pack_reference_order(self) if isinstance(self, P) else None
        if order is not None:
            # a PackedSequence whose ties are not in the reference's order: the result is (see pack_reference_order)
            lens, sorted_indices, unsorted, batch_sizes, bsz_dev, boff = order
            shell = P(data=self.data, batch_sizes=batch_sizes, sorted_indices=sorted_indices, unsorted_indices=unsorted)
            M.adopt_pack(shell, lens, boff, bsz_dev)
            dst = M.lay_pack(shell, lens=lens, boff=boff, n_rows=int(self.data.size(0)))
            return shell._replace(data=O.move(self.data, O.MovePlan(dst, lay, self.data.shape, tmap, arg, name=name)))
        if isinstance(self, P) and tmap == K.T_ROLL:
            # narrow rows: every time step's rows move as ONE run (the ranks that wrap apart) — _meta.lay_pack_steps
            lay = M.lay_pack_steps(self, M.row_bytes(self.data, 1), arg) or lay
        plan = O.MovePlan(lay, lay, self.data.shape, tmap, arg, name=name)
    return self._replace(data=O.move(self.data, plan))


def _cat_roll(self: C, shifts: int) -> C:
    """select/roll.py:6-13: token_ptr' = (t - s + len) % len."""
    return _same_layout_move(self, K.T_ROLL, int(shifts), 'roll')


def _pack_roll(self: P, shifts: int) -> P:
    """select/roll.py:26-30 (3 index-tensor conversions + 1 gather in the reference) as one closed-form
    gather: out[boff[t] + r] = in[boff[(t - s) mod len] + r]; batch_sizes / sorted / unsorted pass through when they
    are what the reference's closing `.pack()` would compute again (core.pack_reference_order), else they are that."""
    return _same_layout_move(self, K.T_ROLL, int(shifts), 'roll')


def _padded_roll(self, shifts: int):
    """select/roll.py:19-23, 33-37.  The reference pads its index tensor with 0, so its padding rows are
    copies of storage row 0; pad_row=0 reproduces that bit for bit."""
    return _same_layout_move(self, K.T_ROLL, int(shifts), 'roll', pad_row=0 if self.data.numel() else -1,
                             logical_stride=True)


C.roll = _cat_roll
P.roll = _pack_roll
L.roll = _padded_roll
R.roll = _padded_roll


def _rev(self: Z):
    """select/rev.py:6-41: per-sequence reversal, t' = len - 1 - t (padding rows = 0 for L/R:
    the reference goes through .left()/.right() with the default fill)."""
    return _same_layout_move(self, K.T_REV_S, 0, 'rev')


for _cls in (C, L, P, R):
    _cls.rev = _rev


# ------------------------------------------------------------------ trunc (select/trunc.py)
def _cat_trunc(self: C, trunc: Tuple[int, int]) -> C:
    """select/trunc.py:9-19."""
    a, b = int(trunc[0]), int(trunc[1])
    B = self.token_sizes.size(0)
    n = int(self.data.size(0)) - B * (a + b)
    dst = M.lay_cat(self.token_sizes, B, n, len_add=-(a + b))
    data = O.move(self.data, O.MovePlan(dst, describe(self), (n,) + _hidden(self), K.T_SHIFT, a, name='trunc'))
    return C(data=data, token_sizes=self.token_sizes - a - b)


def _pack_trunc(self: P, trunc: Tuple[int, int]) -> P:
    """select/trunc.py:38-47: batch_sizes[a+b:], rows (t + a, r)."""
    a, b = int(trunc[0]), int(trunc[1])
    batch_sizes = self.batch_sizes[a + b:]
    n = int(batch_sizes.sum())
    shell = self._replace(batch_sizes=batch_sizes)
    dst = M.lay_pack(shell, lens=M.pack_lens(self), len_add=-(a + b), n_rows=n)
    data = O.move(self.data, O.MovePlan(dst, describe(self), (n,) + _hidden(self), K.T_SHIFT, a, name='trunc'))
    return shell._replace(data=data)


def _padded_trunc(cls):
    def trunc(self, trunc: Tuple[int, int]):
        """select/trunc.py:25-35, 50-62: a view."""
        t = self.size()[1]
        return cls(data=self.data[:, trunc[0]:t - trunc[1]], token_sizes=self.token_sizes - trunc[0] - trunc[1])
    return trunc


C.trunc = _cat_trunc
P.trunc = _pack_trunc
L.trunc = _padded_trunc(L)
R.trunc = _padded_trunc(R)
