"""Kernel launches with autograd: the row mover and the segmented reduce.

Backward of a move is the adjoint move (layouts swapped, token map inverted, zero fill for rows
nothing maps to) — the same kernel.  Backward of a segmented reduce is one fused kernel
(rua_segment_reduce_backward), which also serves scatter_* through the bucket indirection.
"""
import threading
from typing import Callable, Optional, Sequence, Tuple

import torch
from torch import Tensor

from torchrua_amd import _lib as L
from torchrua_amd import _meta as M

# optional hook bench.py installs to bracket named kernels with HIP events on the launch stream
_kernel_hook: Optional[Callable[[str, bool], None]] = None


def set_kernel_hook(fn) -> None:
    global _kernel_hook
    _kernel_hook = fn


_fill_cache = {}


def _fill16(value, dtype: torch.dtype) -> bytes:
    """The fill element replicated to 16 bytes (as the kernel's uint4 pattern); memoised per (value, dtype)."""
    key = (value, dtype) if isinstance(value, (int, float, bool, bytes)) else None
    hit = _fill_cache.get(key) if key is not None else None
    if hit is not None:
        return hit
    raw = value if isinstance(value, bytes) else torch.tensor([value], dtype=dtype).view(torch.uint8).numpy().tobytes()
    if len(raw) > 16:
        raise L.RuaError(f'fill element wider than 16 bytes ({dtype})')
    out = (raw * (16 // len(raw)))[:16]
    if key is not None and len(_fill_cache) < 256:
        _fill_cache[key] = out
    return out


class MovePlan:
    """Everything one rua_move_rows launch needs except the payload pointers."""
    __slots__ = ('dst', 'src', 'tmap', 'arg', 'out_shape', 'fill', 'pad_row', 'flags', 'name')

    def __init__(self, dst: M.Lay, src: M.Lay, out_shape: Sequence[int], tmap: int = L.T_SHIFT, arg: int = 0,
                 fill=0, pad_row: int = -1, flags: int = 0, name: str = 'move'):
        self.dst, self.src, self.tmap, self.arg = dst, src, tmap, arg
        self.out_shape, self.fill, self.pad_row, self.flags, self.name = tuple(out_shape), fill, pad_row, flags, name

    def adjoint(self, src_shape: Sequence[int]) -> 'MovePlan':
        inv = {L.T_SHIFT: (L.T_SHIFT, -self.arg), L.T_ROLL: (L.T_ROLL, -self.arg), L.T_REV_S: (L.T_REV_D, 0),
               L.T_REV_D: (L.T_REV_S, 0)}
        if self.tmap not in inv or self.dst.kind == L.LIST or self.flags:
            raise L.RuaError('this move has no adjoint move')
        tmap, arg = inv[self.tmap]
        return MovePlan(self.src, self.dst, src_shape, tmap, arg, fill=0, name=self.name + '_bwd')


def launch_move(plan: MovePlan, src_data: Tensor, out: Optional[Tensor] = None) -> Tensor:
    dev = L.require_device(src_data)
    lib = L.load()
    src_data = src_data.contiguous()
    if out is None:
        out = torch.empty(plan.out_shape, dtype=src_data.dtype, device=dev)
    elif not out.is_contiguous() or out.dtype != src_data.dtype:
        raise L.RuaError('move target must be contiguous and of the payload dtype')
    # rows are equally wide on both sides; size them on the side the layout `dst` enumerates
    enumerated = src_data if (plan.flags & L.MOVE_SCATTER) else out
    rb = (enumerated.numel() // plan.dst.n_rows) * enumerated.element_size() if plan.dst.n_rows else 0
    fill = _fill16(plan.fill, src_data.dtype)
    if _kernel_hook:
        _kernel_hook(plan.name, True)
    L.check(lib.rua_move_rows(plan.dst.ref(), plan.src.ref(), plan.tmap, plan.arg, L.ptr(out), L.ptr(src_data), rb,
                              fill, plan.pad_row, plan.flags, L.stream_ptr(dev)), 'rua_move_rows')
    if _kernel_hook:
        _kernel_hook(plan.name, False)
    return out


class _Move(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src_data: Tensor, plan: MovePlan):
        ctx.plan = plan
        ctx.src_shape = tuple(src_data.shape)
        return launch_move(plan, src_data)

    @staticmethod
    def backward(ctx, grad: Tensor):
        return _Move.apply(grad.contiguous(), ctx.plan.adjoint(ctx.src_shape)), None


def move(src_data: Tensor, plan: MovePlan) -> Tensor:
    if src_data.requires_grad and torch.is_grad_enabled():
        return _Move.apply(src_data, plan)
    return launch_move(plan, src_data.detach() if src_data.requires_grad else src_data)


class _ListGather(torch.autograd.Function):
    """X[batch_ptr, token_ptr] (core/get.py tuple keys): rows may repeat, so the adjoint accumulates."""

    @staticmethod
    def forward(ctx, src_data: Tensor, plan: MovePlan, flat_fn, lead: int = 1):
        ctx.flat_fn = flat_fn
        ctx.src_shape = tuple(src_data.shape)
        ctx.lead = lead                 # leading dims of src_data that enumerate storage rows (2 for [B, T, *H])
        return launch_move(plan, src_data)

    @staticmethod
    def backward(ctx, grad: Tensor):
        """d/d src = the rows of `grad` summed into the storage rows they were gathered from.  Rows may repeat, so this
        is a scatter-sum: bucket the flat row numbers (stable radix sort, rua_index_buckets) and fold every bucket in
        ascending entry order with the segmented reducer — no float atomics, bitwise reproducible (torch's index_add_
        is neither).  [r4] Twice differentiable, like the reference's `data[key]`: the scatter-sum's own adjoint is the
        gather again (_ScatterSumRows / _GatherRows)."""
        flat = ctx.flat_fn().reshape(-1)
        grad = grad.contiguous()
        hidden = tuple(ctx.src_shape[ctx.lead:])
        n_rows = 1
        for d in ctx.src_shape[:ctx.lead]:
            n_rows *= d
        flat = torch.where(flat < 0, flat + n_rows, flat)          # negative rows wrapped in the forward (like torch)
        g = scatter_sum(grad.reshape((flat.numel(),) + hidden), flat, n_rows)
        return g.reshape(ctx.src_shape), None, None, None


# ------------------------------------------------------------------ reductions
# max / min / logsumexp track the reference's global `initial` through a small scratch (rua.h: `extreme`).  One
# persistent, zeroed scratch per (device, stream): the reduce needs no initialising launch (RUA_OP_SCRATCH_CLEAN) and
# rua_fill_empty hands it back zeroed — stream order makes that safe for one stream, hence the key.
_scratch = {}
# the reduce and its trailing rua_fill_empty share the scratch and must reach the stream back to back: ctypes drops the
# GIL around each call, so a second host thread enqueueing on the SAME stream could slip its own reduce in between
_scratch_pair = threading.Lock()


def extreme_scratch(dev) -> Tuple[Tensor, int]:
    key = (dev.index, torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if dev.index is None else dev.index))
    if torch.cuda.is_current_stream_capturing():
        # inside a graph capture ALWAYS a buffer of the capture's own (zeroed by the legacy initialising launch), even
        # when this stream already has a persistent one: a graph that baked the shared scratch in could be replayed on
        # another stream while eager max / min runs on this one, and the two would race on its flag and ticket words
        return torch.empty(L.EXTREME_WORDS, dtype=torch.long, device=dev), 0
    buf = _scratch.get(key)
    if buf is None:
        buf = _scratch[key] = torch.zeros(L.EXTREME_WORDS, dtype=torch.long, device=dev)
    return buf, L.OP_SCRATCH_CLEAN


def forget_extreme_scratch(dev) -> None:
    for key in [k for k in _scratch if k[0] == dev.index]:
        del _scratch[key]


_EMPTY = {L.SUM: 0.0, L.MEAN: 0.0, L.PROD: 1.0, L.MAX: 0.0, L.MIN: 0.0, L.LOGSUMEXP: float('-inf')}


def _bits(value: float, dtype: torch.dtype) -> int:
    raw = _fill16(value, dtype)[:dtype.itemsize]
    return int.from_bytes(raw, 'little')


def split_workspace(lay: M.Lay, H: int, dtype: torch.dtype, dev, team_ok: bool = True,
                    tail_ok: bool = True) -> Tuple[int, Optional[Tensor]]:
    """(split_rows, workspace) for the reducer's long-sequence splitting (0, None when it is off)."""
    split = M.reduce_split_rows(lay, H * dtype.itemsize, team_ok, tail_ok)
    if not split:
        return 0, None
    nbytes = L.load().rua_reduce_ws_bytes(lay.n_rows, H, L.DTYPES[dtype], split)
    return split, torch.empty(nbytes, dtype=torch.uint8, device=dev)


def int_split_workspace(n_rows: int, H: int, dtype: torch.dtype, dev) -> Tuple[int, Optional[Tensor]]:
    """(split_rows, workspace) of the INTEGER reducer's long-bucket split (rua_reduce_int.hip).  Always armed from a few
    thousand rows on: the bucket sizes live on the device, a skewed histogram is the ordinary case, and when no bucket
    is long the machinery costs two launches of at most ~1 024 waves that find nothing to do."""
    if n_rows < 4096:
        return 0, None
    # about a thousand parts, but no part beyond ~2 MB of payload: a part is ONE wave's walk (profiles/r04_skew.txt)
    split = max(1024, min(n_rows // 1024, (2 << 20) // max(1, H * dtype.itemsize)))
    nbytes = L.load().rua_reduce_ws_bytes(n_rows, H, L.INT_DTYPES[dtype], split)
    return split, torch.empty(nbytes, dtype=torch.uint8, device=dev)


def short_seqs_hint(lay: M.Lay, row_bytes: int) -> int:
    """RUA_OP_SHORT_SEQS for a reduce over a CattedSequence whose lengths the host knows and none of which is far above
    the average (at most 8 x, or 64 rows).  One wave (= one workgroup) per sequence is bound by the workgroup dispatch
    rate when the sequences are short (4 M singletons: 3 ms at any row width; 500 000 sequences of 16 rows: 0.4 ms where
    the payload takes 0.06) and by a chain of dependent loads per sequence when the rows are narrow; with the hint the
    launcher gives every row slot of a wave a sequence of its own up to 16 .. 64 rows on average by row width
    (profiles/r04_cat_ranks_ab.txt).  The wave walks to the longest of its sequences, hence the bound — and no hint at
    all when the lengths live on the device only.  (Four sequences per wave at rows of <= 32 bytes need no hint: that
    form checks its own lengths, wave by wave — and so, since round 5, does every-row-slot-its-own-sequence when the hint
    is withheld: the hint only saves the waves that check.)"""
    if lay.kind != L.CAT or lay.max_len is None or lay.B <= 0 or not 0 < row_bytes <= 512:
        return 0
    avg = lay.n_rows / lay.B
    return L.OP_SHORT_SEQS if lay.max_len <= max(64, 8 * avg) else 0


def launch_reduce(lay: M.Lay, data: Tensor, op: int, out: Optional[Tensor] = None, include_self: int = 0,
                  perm: Optional[Tensor] = None, hidden: Tuple[int, ...] = (), reference_initial: bool = True,
                  name: str = 'reduce', ties_out: Optional[Tensor] = None) -> Tensor:
    """rua_segment_reduce (+ rua_fill_empty for the reference's global-extreme `initial`)."""
    dev = L.require_device(data)
    lib = L.load()
    if data.dtype not in L.DTYPES:
        raise L.RuaError(f'reductions support {list(L.DTYPES)}; got {data.dtype}')
    data = data.contiguous()
    H = 1
    for d in hidden:
        H *= d
    if out is None:
        out = torch.empty((lay.B,) + tuple(hidden), dtype=data.dtype, device=dev)
    elif not out.is_contiguous() or out.dtype != data.dtype or out.numel() != lay.B * H:
        raise L.RuaError('reduce target must be a contiguous [B, *hidden] tensor of the payload dtype')
    extreme, op_bits = None, 0
    if reference_initial and op in (L.MAX, L.MIN, L.LOGSUMEXP) and include_self == 0:
        extreme, op_bits = extreme_scratch(dev)
    tail_ok = include_self != 1 and (data.data_ptr() | out.data_ptr()) % 8 == 0      # the launcher's own condition
    split, ws = split_workspace(lay, H, data.dtype, dev, tail_ok=tail_ok)
    short = short_seqs_hint(lay, H * data.dtype.itemsize) if perm is None and not split else 0
    # (the reduce leaves the global extreme in the scratch — every wave folds the opposite extreme of the rows it reads —
    # so the trailing rua_fill_empty needs no second walk: every workgroup patches its share of the batch, whether or
    # not the host knows how many sequences are empty)
    if _kernel_hook:
        _kernel_hook(name, True)
    paired = extreme is not None
    if paired:
        _scratch_pair.acquire()
    try:
        L.check(lib.rua_segment_reduce(lay.ref(), L.ptr(perm), L.ptr(data), L.ptr(out), H, L.DTYPES[data.dtype],
                                       op | op_bits | short | (L.OP_NO_EMPTY if extreme is not None and lay.no_empty else 0),
                                       include_self,
                                       _bits(_EMPTY[op], data.dtype), L.ptr(extreme), split, L.ptr(ws), L.ptr(ties_out),
                                       L.stream_ptr(dev)), 'rua_segment_reduce')
        if _kernel_hook:
            _kernel_hook(name, False)
        if extreme is not None:
            L.check(lib.rua_fill_empty(lay.ref(), L.ptr(out), H, L.DTYPES[data.dtype], op | (op_bits & L.OP_SCRATCH_CLEAN),
                                       L.ptr(extreme), L.stream_ptr(dev)), 'rua_fill_empty')
    except L.RuaError:
        forget_extreme_scratch(dev)        # a refused launch may have left the flags raised
        raise
    finally:
        if paired:
            _scratch_pair.release()
    return out


class _Reduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data: Tensor, lay: M.Lay, op: int, hidden, lens: Optional[Tensor]):
        ties = None
        if op in (L.MAX, L.MIN):
            # the forward counts, per output element, the elements equal to it (free in the pass that reads the payload
            # anyway): the backward is then ONE walk instead of a counting walk plus an applying walk
            acc = torch.float64 if data.dtype == torch.float64 else torch.float32
            ties = torch.empty((lay.B,) + tuple(hidden), dtype=acc, device=data.device)
        out = launch_reduce(lay, data, op, hidden=hidden, ties_out=ties)
        ctx.lay, ctx.op, ctx.lens, ctx.ties = lay, op, lens, ties
        ctx.save_for_backward(data.contiguous(), out)
        return out

    @staticmethod
    def backward(ctx, grad: Tensor):
        data, out = ctx.saved_tensors
        if torch.is_grad_enabled() and ctx.op in (L.MAX, L.MIN, L.LOGSUMEXP):
            # [r5] a graph of this backward is being recorded (create_graph=True).  The reference's reductions are ATen
            # compositions and differentiate any number of times (reduce.py:34-61: torch.segment_reduce; logsumexp = detached
            # max, exp, segment sum, log); the fused kernel's output carries no graph, so HERE the gradient is spelled with
            # differentiable pieces: the per-sequence broadcast (the backward of sum, whose own adjoint is the reduction)
            # and torch's elementwise ops — [N, H] temporaries, paid only by callers who ask for second derivatives
            return _composed_reduce_grad(grad, data, out, ctx.lay, ctx.op, ctx.ties), None, None, None, None
        return _ReduceBwd.apply(grad, data, out, ctx.lay, ctx.op, ctx.ties), None, None, None, None


def _composed_reduce_grad(grad: Tensor, data: Tensor, out: Tensor, lay: M.Lay, op: int, ties: Optional[Tensor]) -> Tensor:
    """d reduce / d data as a differentiable function of (grad, data, out): max / min / logsumexp under create_graph."""
    def spread(v: Tensor) -> Tensor:                 # every sequence's row of `v` over the sequence's storage rows
        return _ReduceBwd.apply(v.contiguous(), data.detach(), out.detach(), lay, L.SUM, None)

    if op == L.LOGSUMEXP:
        return spread(grad) * (data - spread(out)).exp()
    o = spread(out.detach())
    x = data.detach()
    hit = (x == o) | ((x != x) & (o != o))
    g32 = grad.to(ties.dtype)
    share = torch.where(g32 > 0, g32 / ties.clamp_min(1), g32).to(grad.dtype)     # torch.segment_reduce's tie rule
    return spread(share) * hit


class _ReduceBwd(torch.autograd.Function):
    """The backward of a segmented reduce as a function of the cotangent.  One fused kernel
    (rua_segment_reduce_backward): reads the payload once (max/min: the forward already counted the ties), writes the
    gradient once — no [N, H] temporaries.  [r4] For SUM and MEAN the map cotangent -> gradient is linear (a broadcast
    of every segment's row over the segment's rows, divided by the length for MEAN) and its adjoint is the reduction
    itself, so these two are differentiable any number of times, like the reference's (torch.segment_reduce,
    reduce.py:44-49).  [r5] max / min / logsumexp: under create_graph=True _Reduce.backward composes the gradient from
    differentiable pieces instead of calling this kernel (_composed_reduce_grad), so they are twice differentiable too;
    prod differentiates once (the reference's does not: a stated gap)."""

    @staticmethod
    def forward(ctx, grad: Tensor, data: Tensor, out: Tensor, lay: M.Lay, op: int, ties: Optional[Tensor]):
        dev = L.require_device(data)
        lib = L.load()
        grad = grad.contiguous()
        g = torch.empty(data.shape, dtype=data.dtype, device=dev)   # padding rows: zeroed by the call (BWD_FILL_PADDING)
        H = 1
        for d in out.shape[1:]:
            H *= d
        split, ws = split_workspace(lay, H, data.dtype, dev, team_ok=False)      # (the backward walk has no wave teams)
        # max/min: `ties` counted by the forward -> apply only (TIES_FINAL).  segment_max/min are torch.segment_reduce in
        # the reference (reduce.py:34-41), whose backward lets tied extrema share a positive gradient and hands each of
        # them a non-positive one whole (BWD_TIES_POSITIVE)
        L.check(lib.rua_segment_reduce_backward(lay.ref(), None, L.ptr(data), L.ptr(out), L.ptr(grad), L.ptr(g), H,
                                                L.DTYPES[data.dtype], op,
                                                (L.TIES_FINAL if ties is not None else 0) | L.BWD_FILL_PADDING | L.BWD_TIES_POSITIVE,
                                                split, L.ptr(ws), L.ptr(ties), None, L.stream_ptr(dev)),
                'rua_segment_reduce_backward')
        ctx.lay, ctx.op, ctx.hidden = lay, op, tuple(out.shape[1:])
        return g

    @staticmethod
    def backward(ctx, gg: Tensor):
        if ctx.op not in (L.SUM, L.MEAN):
            raise RuntimeError('torchrua_amd: prod (and the fused max / min / logsumexp kernel outside create_graph) '
                               'differentiate once; second-order gradients exist for sum, mean, max, min and logsumexp '
                               '(and for every cast, select and gather)')
        return reduce(gg.contiguous(), ctx.lay, ctx.op, ctx.hidden, None), None, None, None, None, None


def reduce(data: Tensor, lay: M.Lay, op: int, hidden, lens: Optional[Tensor]) -> Tensor:
    if data.requires_grad and torch.is_grad_enabled():
        return _Reduce.apply(data, lay, op, tuple(hidden), lens)
    return launch_reduce(lay, data.detach() if data.requires_grad else data, op, hidden=tuple(hidden))


# ------------------------------------------------------------------ scatter-sum of rows (adjoint of a row gather)
def index_buckets(index: Tensor, S: int) -> Tuple[Tensor, Tensor]:
    """(counts[S], perm[M]): the entries of `index` bucketed by destination, every bucket in ascending entry order
    (rua_index_buckets: a stable LSD radix sort, deterministic for any fan-in)."""
    dev = L.require_device(index)
    lib = L.load()
    index = M._as_lens(index)
    m = index.numel()
    counts = torch.empty(S, dtype=torch.long, device=dev)
    off = torch.empty(S, dtype=torch.long, device=dev)
    perm = torch.empty(m, dtype=torch.long, device=dev)
    ws = torch.empty(lib.rua_bucket_ws_elems(m, S), dtype=torch.long, device=dev)
    L.check(lib.rua_index_buckets(L.ptr(index), m, S, L.ptr(counts), L.ptr(off), L.ptr(perm), L.ptr(ws),
                                  L.stream_ptr(dev)), 'rua_index_buckets')
    M._memo_put(counts, 'off', off)
    return counts, perm


class _ScatterSumRows(torch.autograd.Function):
    """out[s] = sum of rows[i] over index[i] == s; adjoint: the gather rows[index]."""

    @staticmethod
    def forward(ctx, rows: Tensor, index: Tensor, n_out: int):
        ctx.save_for_backward(index)        # (saved, not a plain attribute: autograd then checks its version)
        return scatter_sum_rows(rows.detach(), index, n_out)

    @staticmethod
    def backward(ctx, g: Tensor):
        index, = ctx.saved_tensors
        return gather_rows(g.contiguous(), index), None, None


class _GatherRows(torch.autograd.Function):
    """out[i] = rows[index[i]] (zeros where index[i] is out of range); adjoint: the scatter-sum."""

    @staticmethod
    def forward(ctx, rows: Tensor, index: Tensor):
        ctx.save_for_backward(index)
        ctx.n = int(rows.size(0))
        plan = MovePlan(M.lay_list(None, index), M.lay_flat(ctx.n), (index.numel(),) + tuple(rows.shape[1:]), fill=0,
                        name='gather_rows')
        return launch_move(plan, rows.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        index, = ctx.saved_tensors
        return scatter_sum(g.contiguous(), index, ctx.n), None


def gather_rows(rows: Tensor, index: Tensor) -> Tensor:
    """rows[index] through the mover; recorded by autograd when `rows` carries a graph (second-order gradients)."""
    if rows.requires_grad and torch.is_grad_enabled():
        return _GatherRows.apply(rows, index)
    return _GatherRows.forward(_NoCtx(), rows, index)


def scatter_sum(rows: Tensor, index: Tensor, n_out: int) -> Tensor:
    """scatter_sum_rows, recorded by autograd when `rows` carries a graph (second-order gradients)."""
    if rows.requires_grad and torch.is_grad_enabled():
        return _ScatterSumRows.apply(rows, index, n_out)
    return scatter_sum_rows(rows, index, n_out)


class _NoCtx:
    """Stands in for an autograd context when a Function's forward is run for its value only."""

    def save_for_backward(self, *tensors) -> None:
        pass


def scatter_sum_rows(rows: Tensor, index: Tensor, n_out: int) -> Tensor:
    """out[s] = sum of rows[i] over the entries i with index[i] == s, s < n_out (rows nobody names are 0)."""
    hidden = tuple(rows.shape[1:])
    counts, perm = index_buckets(index, n_out)
    lay = M.lay_cat(counts, n_out, int(rows.size(0)))
    lay.heavy_tail = True
    if rows.dtype in L.INT_DTYPES:        # (integer payloads carry no gradient: the integer reducer, for completeness)
        out = torch.empty((n_out,) + hidden, dtype=rows.dtype, device=rows.device)
        H = 1
        for d in hidden:
            H *= d
        split, ws = int_split_workspace(int(rows.size(0)), H, rows.dtype, rows.device)
        L.check(L.load().rua_segment_reduce(lay.ref(), L.ptr(perm), L.ptr(rows.contiguous()), L.ptr(out), H,
                                            L.INT_DTYPES[rows.dtype], L.SUM, 0, 0, None, split, L.ptr(ws), None,
                                            L.stream_ptr(rows.device)), 'rua_segment_reduce')
        return out
    return launch_reduce(lay, rows, L.SUM, perm=perm, hidden=hidden, reference_initial=False, name='scatter')
