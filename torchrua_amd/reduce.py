"""Segmented and scatter reductions — mirror of torchrua.reduce (reference reduce.py).

segment_*  : torch.segment_reduce(..., lengths=sizes, unsafe=True, initial=..) in the reference
             (reduce.py:34-61) -> rua_segment_reduce over a CAT layout (fp32 accumulation, one pass;
             the reference's extra full read for `initial = tensor.min()` is folded into that pass).
scatter_*  : torch.index_reduce / index_add (reduce.py:6-31) -> bucket the index (stable radix sort,
             rua_index_buckets) and run the same segmented kernel through the row indirection:
             no float atomics, bitwise reproducible for any fan-in.
reduce_*   : the same reductions over the sequences of ANY container (C/L/P/R) -> [B, *H] in batch
             order, without first converting to C (BASELINE.json's `reduce_sum`; SURVEY.md §8d spells
             it in the reference as p.cat() + segment_sum, or scatter_sum over p.ptr()[0]).
"""
import os

import torch
from torch.autograd.function import once_differentiable

from torchrua_amd import _lib as K
from torchrua_amd import _meta as M
from torchrua_amd import _ops as O
from torchrua_amd.layout import C, P, T, Z, describe, lens_of

__all__ = [
    'segment_max', 'segment_min', 'segment_sum', 'segment_mean', 'segment_prod', 'segment_logsumexp',
    'segment_head', 'segment_last',
    'scatter_max', 'scatter_min', 'scatter_sum', 'scatter_mean', 'scatter_prod', 'scatter_logsumexp',
    'reduce_max', 'reduce_min', 'reduce_sum', 'reduce_mean', 'reduce_prod', 'reduce_logsumexp',
    'pack_reduce',
]


def _segment(tensor: T, segment_sizes: T, op: int) -> T:
    K.require_device(tensor, segment_sizes)
    S = segment_sizes.numel()
    lay = M.lay_cat(segment_sizes, S, int(tensor.size(0)))
    return O.reduce(tensor, lay, op, tuple(tensor.shape[1:]), segment_sizes)


def segment_max(tensor: T, segment_sizes: T) -> T:
    """reduce.py:34-36 (empty segment -> tensor.min(), as the reference's `initial`)."""
    return _segment(tensor, segment_sizes, K.MAX)


def segment_min(tensor: T, segment_sizes: T) -> T:
    """reduce.py:39-41."""
    return _segment(tensor, segment_sizes, K.MIN)


def segment_sum(tensor: T, segment_sizes: T) -> T:
    """reduce.py:44-45."""
    return _segment(tensor, segment_sizes, K.SUM)


def segment_mean(tensor: T, segment_sizes: T) -> T:
    """reduce.py:48-49."""
    return _segment(tensor, segment_sizes, K.MEAN)


def segment_prod(tensor: T, segment_sizes: T) -> T:
    """reduce.py:52-53."""
    return _segment(tensor, segment_sizes, K.PROD)


def segment_logsumexp(tensor: T, segment_sizes: T) -> T:
    """reduce.py:56-61 (max, sub, exp, sum, log over [N,H] temporaries) as one online pass."""
    return _segment(tensor, segment_sizes, K.LOGSUMEXP)


def segment_head(tensor: T, segment_sizes: T) -> T:
    """reduce.py:64-65."""
    return C(data=tensor, token_sizes=segment_sizes).head(n=1).data


def segment_last(tensor: T, segment_sizes: T) -> T:
    """reduce.py:68-69."""
    return C(data=tensor, token_sizes=segment_sizes).last()


# ------------------------------------------------------------------ reduce over sequences of any layout
def _reduce_seq(sequence: Z, op: int) -> T:
    hidden = tuple(sequence.data.shape[1:]) if isinstance(sequence, (C, P)) else tuple(sequence.data.shape[2:])
    return O.reduce(sequence.data, describe(sequence), op, hidden, lens_of(sequence))


def reduce_sum(sequence: Z) -> T:
    return _reduce_seq(sequence, K.SUM)


def reduce_mean(sequence: Z) -> T:
    return _reduce_seq(sequence, K.MEAN)


def reduce_max(sequence: Z) -> T:
    return _reduce_seq(sequence, K.MAX)


def reduce_min(sequence: Z) -> T:
    return _reduce_seq(sequence, K.MIN)


def reduce_prod(sequence: Z) -> T:
    return _reduce_seq(sequence, K.PROD)


def reduce_logsumexp(sequence: Z) -> T:
    return _reduce_seq(sequence, K.LOGSUMEXP)


# ------------------------------------------------------------------ scatter_* (reduce.py:6-31)
_buckets = O.index_buckets


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensor: T, index: T, source: T, op: int, include_self: bool):
        S = tensor.size(0)
        counts, perm = _buckets(index, S)
        lay = M.lay_cat(counts, S, int(source.size(0)))
        lay.heavy_tail = True
        hidden = tuple(tensor.shape[1:])
        # (row-major whatever the strides of `tensor`: the kernel addresses out[s * H + h])
        if op == K.SUM and not include_self:
            out, mode = torch.empty(tensor.shape, dtype=tensor.dtype, device=tensor.device), 0   # index_add into zeros (reduce.py:15)
        else:
            out = tensor.detach().clone(memory_format=torch.contiguous_format)
            # include_self: fold the old row in; otherwise rows no index names keep their value
            # and touched rows start from the identity (torch.index_reduce semantics)
            mode = 1 if include_self else 2
        if op == K.LOGSUMEXP and not include_self:
            out, mode = torch.empty(tensor.shape, dtype=tensor.dtype, device=tensor.device), 0   # untouched rows -> log(0) = -inf (reduce.py:26-31)
        ties = None
        if op in (K.MAX, K.MIN) and (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]):
            # source rows equal to the result, counted by the forward (the backward is then one walk)
            ties = torch.zeros(out.shape, dtype=torch.float64 if source.dtype == torch.float64 else torch.float32,
                               device=source.device)
        O.launch_reduce(lay, source.detach(), op, out=out, include_self=mode, perm=perm, hidden=hidden,
                        reference_initial=False, name='scatter', ties_out=ties)
        ctx.op, ctx.include_self, ctx.lay, ctx.perm, ctx.ties = op, include_self, lay, perm, ties
        ctx.save_for_backward(tensor, index, source, out, counts)
        return out

    @staticmethod
    def backward(ctx, grad: T):
        if ctx.op == K.SUM and torch.is_grad_enabled() and grad.requires_grad:
            # [r4] a graph of the backward is being recorded (create_graph=True): scatter_sum is linear, its gradient
            # w.r.t. the source rows is the gather grad[index] — spelled with the differentiable gather, whose adjoint is
            # the scatter-sum again (the reference's index_add is twice differentiable the same way)
            tensor, index, source, out, counts = ctx.saved_tensors
            g_src = O.gather_rows(grad.contiguous(), M._as_lens(index)) if ctx.needs_input_grad[2] else None
            g_ten = grad if (ctx.include_self and ctx.needs_input_grad[0]) else None
            return g_ten, None, g_src, None, None
        return _Scatter._backward_once(ctx, grad)

    @staticmethod
    def _backward_impl(ctx, grad: T):
        """Gradient w.r.t. the source rows: the fused kernel walking every destination's bucket
        (rua_segment_reduce_backward with the row indirection; the old destination row `tensor` rides along as
        `self_in`: one more tie candidate for max/min, one more factor for prod with include_self).  Gradient w.r.t.
        `tensor`: one elementwise launch over [S, H] (rua_scatter_self_grad).  No ATen kernel touches either."""
        tensor, index, source, out, counts = ctx.saved_tensors
        op, inc = ctx.op, ctx.include_self
        lib = K.load()
        dev = K.require_device(source)
        dt = K.DTYPES[source.dtype]
        grad = grad.contiguous()
        tensor_c = tensor.contiguous()
        hidden = tuple(out.shape[1:])
        S, H = out.size(0), 1
        for d in hidden:
            H *= d
        g_src = g_ten = None
        if ctx.needs_input_grad[2]:
            # every source row sits in one bucket — unless its index is out of range: the bucket builder drops such
            # entries (torch's index_add / index_reduce device-assert on them), and their gradient rows would stay
            # unwritten.  Small gradients are simply zeroed first; large ones are not (a second pass over N x H), and
            # RUA_CHECK_INDEX=1 turns the silent drop into torch's IndexError in the forward (_scatter)
            g_src = (torch.zeros if source.numel() * source.element_size() <= _ZERO_GRAD_BYTES else torch.empty)(
                source.shape, dtype=source.dtype, device=dev)
            ties, mode, self_in = None, (1 if inc else 0), None
            if op in (K.MAX, K.MIN):
                # the forward counted the source rows equal to the result; the kernel adds the old row's tie
                # (torch counts it with and without include_self: index_reduce_backward) -> one walk
                ties, mode, self_in = ctx.ties, K.TIES_FINAL, tensor_c
            elif op == K.PROD and inc:
                self_in = tensor_c
            split, ws = O.split_workspace(ctx.lay, H, source.dtype, dev, team_ok=False)
            K.check(lib.rua_segment_reduce_backward(ctx.lay.ref(), K.ptr(ctx.perm), K.ptr(source.contiguous()),
                                                    K.ptr(out), K.ptr(grad), K.ptr(g_src), H, dt, op, mode, split,
                                                    K.ptr(ws), K.ptr(ties), K.ptr(self_in), K.stream_ptr(dev)),
                    'rua_segment_reduce_backward')
        if ctx.needs_input_grad[0]:
            if op == K.SUM:
                g_ten = grad if inc else None            # index_add: d out / d tensor = 1 (reduce.py:14-15 adds into zeros)
            elif op == K.LOGSUMEXP and not inc:
                g_ten = None                             # reduce.py:26-31 scatters into a fresh buffer
            else:
                aux = None
                if op in (K.MAX, K.MIN) and inc:
                    aux = ctx.ties
                elif op == K.PROD and inc:
                    # d out / d tensor = product of the bucket's source rows.  `out / tensor` loses it where
                    # tensor == 0 (torch masks those and reduces again): one more launch of the reducer, into ones
                    aux = torch.ones(out.shape, dtype=out.dtype, device=dev)
                    O.launch_reduce(ctx.lay, source.detach(), K.PROD, out=aux, include_self=2, perm=ctx.perm,
                                    hidden=hidden, reference_initial=False, name='scatter')
                g_ten = torch.empty(out.shape, dtype=out.dtype, device=dev)
                K.check(lib.rua_scatter_self_grad(K.ptr(counts), S, H, K.ptr(tensor_c), K.ptr(out), K.ptr(grad),
                                                  K.ptr(aux), K.ptr(g_ten), dt, op, 1 if inc else 0,
                                                  K.stream_ptr(dev)), 'rua_scatter_self_grad')
        return g_ten, None, g_src, None, None


_Scatter._backward_once = staticmethod(once_differentiable(_Scatter._backward_impl))      # every op but the one above


_ZERO_GRAD_BYTES = 16 << 20
_CHECK_INDEX = os.environ.get('RUA_CHECK_INDEX', '') not in ('', '0')


def _scatter(tensor: T, index: T, source: T, op: int, include_self: bool, dim: int) -> T:
    K.require_device(tensor, index, source)
    if dim != 0:
        # the kernels reduce along the row dimension; any other `dim` (torch.index_reduce accepts one, the reference
        # passes it through: reduce.py:6-31) is brought to the front and back — a strided view in, one contiguous
        # copy inside, a view out; gradients follow the views
        nd = tensor.dim()
        if not -nd <= dim < nd or source.dim() != nd:
            raise K.RuaError(f'scatter_*: dim {dim} out of range for a {nd}-d tensor')
        d = dim % nd
        out = _scatter(tensor.movedim(d, 0), index, source.movedim(d, 0), op, include_self, 0)
        return out.movedim(0, d)
    if tensor.dtype not in K.DTYPES and tensor.dtype not in K.INT_DTYPES:
        raise K.RuaError(f'scatter_* supports {list(K.DTYPES) + list(K.INT_DTYPES)}; got {tensor.dtype}')
    # what torch.index_add / index_reduce reject (reduce.py:6-31 inherit their checks): the kernel is launched with
    # the source's element type and the destination's row width, so a mismatch would write out of bounds
    if source.dtype != tensor.dtype:
        raise K.RuaError(f'scatter_*: source is {source.dtype} but tensor is {tensor.dtype}')
    if tensor.dim() < 1 or source.dim() != tensor.dim() or tuple(source.shape[1:]) != tuple(tensor.shape[1:]):
        raise K.RuaError(f'scatter_*: source rows {tuple(source.shape[1:])} do not match tensor rows '
                         f'{tuple(tensor.shape[1:])}')
    if index.dim() != 1 or index.numel() != source.size(0):
        raise K.RuaError(f'scatter_*: index must be 1-D with one entry per source row '
                         f'(got {tuple(index.shape)} for {source.size(0)} rows)')
    if index.dtype not in (torch.long, torch.int32):
        raise K.RuaError(f'scatter_*: index must be int64 or int32 (got {index.dtype})')
    if _CHECK_INDEX and index.numel():
        # debug aid (one sync): entries outside [0, S) are DROPPED by the bucket builder (rua.h: rua_index_buckets),
        # where torch.index_add / index_reduce device-assert; with the switch on they raise like torch's CPU path
        lo, hi = int(index.min()), int(index.max())
        if lo < 0 or hi >= tensor.size(0):
            raise IndexError(f'scatter_*: index out of range [0, {tensor.size(0)}): min {lo}, max {hi}')
    if tensor.dtype in K.INT_DTYPES:
        return _scatter_int(tensor, index, source, op, bool(include_self))
    return _Scatter.apply(tensor, index, source, op, bool(include_self))


def _scatter_int(tensor: T, index: T, source: T, op: int, include_self: bool) -> T:
    """scatter_* on integer tensors: the reference hands them to torch.index_reduce / index_add like any other dtype
    (reduce.py:6-23).  Same buckets, an integer reducer (rua_reduce_int.hip): the element type's own wrapping sums and
    products, and ATen's floor division by a count of that type for `mean`.  Bit-exact; integers carry no gradient."""
    if op == K.LOGSUMEXP:
        # reduce.py:26-31 on integers: the maximum and the two differences are taken in the integer type, `.exp()`
        # promotes to float32 and the result is float32.  [r5] The same composition over this library's own scatter_max
        # (integer kernel, bit-exact) and scatter_sum (float kernel); the four elementwise steps on [S, H] / [M, H] are
        # torch's, as in the reference.  For uint8 the differences wrap to 256 - d and the reference's own answer is
        # inf wherever a bucket holds two different values (tests/golden/META.json): that stays an error here.
        if tensor.dtype == torch.uint8:
            raise K.RuaError('scatter_logsumexp of uint8 tensors: the reference\'s differences wrap (its result is inf); '
                             'convert to a signed or floating type')
        m = _scatter_int(tensor, index, source, K.MAX, include_self)
        t = (tensor - m).exp()
        s = (source - m.index_select(0, index.long())).exp()
        return _Scatter.apply(t, index, s, K.SUM, include_self).log() + m
    S = tensor.size(0)
    counts, perm = _buckets(index, S)
    lay = M.lay_cat(counts, S, int(source.size(0)))
    H = 1
    for d in tensor.shape[1:]:
        H *= d
    if op == K.SUM and not include_self:
        out, mode = torch.empty(tensor.shape, dtype=tensor.dtype, device=tensor.device), 0      # index_add into zeros
    else:
        out, mode = tensor.detach().clone(memory_format=torch.contiguous_format), (1 if include_self else 2)
    source = source.detach().contiguous()
    split, ws = O.int_split_workspace(int(source.size(0)), H, tensor.dtype, tensor.device)
    K.check(K.load().rua_segment_reduce(lay.ref(), K.ptr(perm), K.ptr(source), K.ptr(out), H, K.INT_DTYPES[tensor.dtype],
                                        op, mode, 0, None, split, K.ptr(ws), None, K.stream_ptr(tensor.device)),
            'rua_segment_reduce')
    return out


def scatter_max(tensor: T, index: T, source: T, include_self: bool = False, dim: int = 0):
    """reduce.py:6-7."""
    return _scatter(tensor, index, source, K.MAX, include_self, dim)


def scatter_min(tensor: T, index: T, source: T, include_self: bool = False, dim: int = 0):
    """reduce.py:10-11."""
    return _scatter(tensor, index, source, K.MIN, include_self, dim)


def scatter_sum(tensor: T, index: T, source: T, include_self: bool = False, dim: int = 0):
    """reduce.py:14-15."""
    return _scatter(tensor, index, source, K.SUM, include_self, dim)


def scatter_mean(tensor: T, index: T, source: T, include_self: bool = False, dim: int = 0):
    """reduce.py:18-19."""
    return _scatter(tensor, index, source, K.MEAN, include_self, dim)


def scatter_prod(tensor: T, index: T, source: T, include_self: bool = False, dim: int = 0):
    """reduce.py:22-23."""
    return _scatter(tensor, index, source, K.PROD, include_self, dim)


def scatter_logsumexp(tensor: T, index: T, source: T, include_self: bool = False, dim: int = 0):
    """reduce.py:26-31."""
    return _scatter(tensor, index, source, K.LOGSUMEXP, include_self, dim)


# ------------------------------------------------------------------ fused pack + reduce (extension)
_OPS = {'sum': K.SUM, 'mean': K.MEAN, 'max': K.MAX, 'min': K.MIN, 'prod': K.PROD, 'logsumexp': K.LOGSUMEXP}


FUSED_MIN_UNITS = 16384      # below this many (sequence, column chunk) units pack_reduce takes the two-kernel form


def pack_reduce(sequence: Z, op: str = 'sum', fused: bool = None):
    """(sequence.pack(), reduce_<op>(that PackedSequence)) in ONE pass over the payload.

    An extension with no one-call twin in the reference: it equals core/cast.py:41-49 followed by the
    per-sequence reduction (reduce.py:34-61 spelled over the packed rows) — the PackedSequence bit for bit, the
    reduction in the same fp32 arithmetic (another association of the partial sums only where reduce_* shares a
    sequence among a team of waves) — but reads the
    payload once instead of twice (2*N*H*e + B*H*e bytes instead of 3*N*H*e + B*H*e).  Falls back to the
    two-kernel form when autograd is recording, for a PackedSequence input, or for rows that are not a
    multiple of 16 bytes."""
    from torchrua_amd.core import _hidden, _pack_meta
    code = _OPS[op]
    data = sequence.data
    hidden = _hidden(sequence)
    H = 1
    for d in hidden:
        H *= d
    fusable = (not isinstance(sequence, P) and data.dtype in K.DTYPES and data.is_contiguous()
               and (H * data.element_size()) % 16 == 0 and data.data_ptr() % 16 == 0
               and not (data.requires_grad and torch.is_grad_enabled()))
    # the fused kernel gives one wave to each (sequence, column chunk): with few units it cannot fill the chip, while
    # the two-kernel form moves rows at tile granularity and reduces with teams of waves (B = 4 096, H = 256: 0.93 ms
    # fused against 0.29 ms for pack + reduce)
    n_seq = int(sequence.token_sizes.numel()) if not isinstance(sequence, P) else 0
    row_bytes = H * data.element_size()
    n_chunks = 1 if row_bytes <= 1024 else -(-row_bytes // 4096)
    if fused is False or (fused is None and n_seq * n_chunks < FUSED_MIN_UNITS):     # fused=True: whenever it can
        fusable = False
    if not fusable:
        p = sequence.pack()
        return p, _reduce_seq(p, code)
    dev = K.require_device(data)
    lib = K.load()
    lens, sorted_indices, unsorted, batch_sizes, bsz_dev, boff = _pack_meta(sequence.token_sizes, dev)
    n = int(data.size(0)) if isinstance(sequence, C) else M.total_len(sequence.token_sizes)
    B = lens.numel()
    pdata = torch.empty((n,) + hidden, dtype=data.dtype, device=dev)
    p = P(data=pdata, batch_sizes=batch_sizes, sorted_indices=sorted_indices, unsorted_indices=unsorted)
    M.adopt_pack(p, lens, boff, bsz_dev)
    dst = M.lay_pack(p, lens=lens, boff=boff, T=batch_sizes.numel(), n_rows=n)
    src = describe(sequence)
    out = torch.empty((B,) + hidden, dtype=data.dtype, device=dev)
    extreme, op_bits = O.extreme_scratch(dev) if code in (K.MAX, K.MIN, K.LOGSUMEXP) else (None, 0)
    split, ws = O.split_workspace(src, H, data.dtype, dev, team_ok=False)      # (the fused kernel has no wave teams)
    if O._kernel_hook:
        O._kernel_hook('pack_reduce', True)
    paired = extreme is not None          # as in _ops.launch_reduce: the two launches stay back to back on the stream
    if paired:
        O._scratch_pair.acquire()
    try:
        K.check(lib.rua_pack_reduce(src.ref(), dst.ref(), K.ptr(data), K.ptr(pdata), K.ptr(out), H, K.DTYPES[data.dtype],
                                    code | op_bits,
                                    O._bits(O._EMPTY[code], data.dtype), K.ptr(extreme), split, K.ptr(ws),
                                    K.stream_ptr(dev)), 'rua_pack_reduce')
        if O._kernel_hook:
            O._kernel_hook('pack_reduce', False)
        if extreme is not None:
            K.check(lib.rua_fill_empty(src.ref(), K.ptr(out), H, K.DTYPES[data.dtype], code | (op_bits & K.OP_SCRATCH_CLEAN),
                                       K.ptr(extreme), K.stream_ptr(dev)), 'rua_fill_empty')
    except K.RuaError:
        O.forget_extreme_scratch(dev)
        raise
    finally:
        if paired:
            O._scratch_pair.release()
    return p, out
