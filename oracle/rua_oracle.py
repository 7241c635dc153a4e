"""CPU oracle: a numpy + C restatement of the reference's algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg — as the checker / the timed CPU baseline, never by the product package
(torchrua_amd has no CPU path and never imports this module).

It follows the reference (speedcell4/torchrua 0.5.1, /root/reference) step by step — build the
(batch_ptr, token_ptr) enumeration, turn it into flat row indices, gather / scatter — rather than
the closed-form row maps the HIP kernels use, so the two are independent derivations of the same
contract.  Every function cites the reference lines it restates.  The arithmetic the reference
delegates to the third-party module `torch` (unpinned: pyproject.toml:7-9; 2.10.0 here) is
restated in oracle/rua_oracle.c.  Parity pin: tests/golden/*.npz, produced by running the
reference itself (oracle/gen_golden.py); tests/test_oracle_golden.py checks this module against
every vector.

Payloads are numpy arrays [rows, *H] of any dtype (bf16 travels as its uint16 bit pattern for the
copy ops; reductions take float32/float64).  Index vectors are int64 like the reference's.
`sorted_indices` is an INPUT of `pack_view`: the reference obtains it from a host
`torch.sort(descending=True)` whose tie order is implementation-defined (SURVEY.md §8a note).
"""
import ctypes
import os
from dataclasses import dataclass, replace
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'librua_oracle.so')
_lib = None

I64 = np.int64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f'{_LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())')
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=I64)


# --------------------------------------------------------------------------- utils.py
def get_offsets(sizes):
    """utils.py:16-19."""
    sizes = _i64(sizes)
    out = np.empty_like(sizes)
    lib().orc_get_offsets(_p(sizes), ctypes.c_int64(sizes.size), _p(out))
    return out


def major_sizes_to_ptr(sizes):
    """utils.py:7-13 -> (major_ptr, minor_ptr)."""
    sizes = _i64(sizes)
    n = int(sizes.sum())
    major, minor = np.empty(n, I64), np.empty(n, I64)
    lib().orc_major_sizes_to_ptr(_p(sizes), ctypes.c_int64(sizes.size), _p(major), _p(minor))
    return major, minor


def invert_permutation(p):
    """utils.py:22-26."""
    p = _i64(p)
    inv = np.empty_like(p)
    lib().orc_invert_permutation(_p(p), ctypes.c_int64(p.size), _p(inv))
    return inv


def _gather(data, idx):
    """`data[key]` — core/get.py:29,42,61,74."""
    data = np.ascontiguousarray(data)
    idx = _i64(idx)
    idx = np.where(idx < 0, idx + data.shape[0], idx)        # aten::index wraps negative indices
    assert idx.size == 0 or (idx.min() >= 0 and idx.max() < data.shape[0]), 'index out of range'
    out = np.empty(idx.shape + data.shape[1:], data.dtype)   # index tensors keep their shape
    rb = int(np.prod(data.shape[1:], dtype=np.int64)) * data.itemsize
    if idx.size and rb:
        lib().orc_gather_rows(_p(data), _p(idx), ctypes.c_int64(idx.size), ctypes.c_int64(rb), _p(out))
    return out


def _scatter(dst, idx, src):
    """`data[key] = value` — core/set.py:30,45,67,82 (dst modified in place; must be contiguous)."""
    assert dst.flags.c_contiguous
    idx = _i64(idx)
    idx = np.where(idx < 0, idx + dst.shape[0], idx)
    assert idx.size == 0 or (idx.min() >= 0 and idx.max() < dst.shape[0]), 'index out of range'
    src = np.ascontiguousarray(np.broadcast_to(np.asarray(src, dtype=dst.dtype), idx.shape + dst.shape[1:]))   # index_put_ broadcasts the value
    rb = int(np.prod(dst.shape[1:], dtype=np.int64)) * dst.itemsize
    if idx.size and rb:
        lib().orc_scatter_rows(_p(dst), _p(idx), ctypes.c_int64(idx.size), ctypes.c_int64(rb), _p(src))


def _full(shape, fill, dtype):
    """data.new_full — core/view.py:36,69."""
    out = np.empty(shape, dtype)
    pat = np.array([fill]).astype(dtype) if not isinstance(fill, np.ndarray) else fill.astype(dtype).reshape(1)
    if out.size:
        lib().orc_fill(_p(out), ctypes.c_int64(out.size), ctypes.c_int64(out.itemsize), _p(pat))
    return out


# --------------------------------------------------------------------------- layout/*.py
@dataclass
class Seq:
    """One of the reference's four containers.  kind 'C' layout/cat.py:9-11, 'L' left.py:9-11,
    'R' right.py:10-12 (fields data, token_sizes); 'P' = torch PackedSequence, pack.py:5-9
    (fields data, batch_sizes, sorted_indices, unsorted_indices)."""
    kind: str
    data: np.ndarray
    token_sizes: Optional[np.ndarray] = None
    batch_sizes: Optional[np.ndarray] = None
    sorted_indices: Optional[np.ndarray] = None
    unsorted_indices: Optional[np.ndarray] = None

    def with_data(self, data):
        return replace(self, data=data)


def C(data, token_sizes):
    return Seq('C', data, token_sizes=_i64(token_sizes))


def L(data, token_sizes):
    return Seq('L', data, token_sizes=_i64(token_sizes))


def R(data, token_sizes):
    return Seq('R', data, token_sizes=_i64(token_sizes))


def P(data, batch_sizes, sorted_indices, unsorted_indices):
    return Seq('P', data, batch_sizes=_i64(batch_sizes), sorted_indices=_i64(sorted_indices),
               unsorted_indices=_i64(unsorted_indices))


def size(s: Seq) -> Tuple[int, ...]:
    """cat.py:61-66, left.py:61-66, right.py:62-67, pack.py:12-17 -> (b, t, *hidden)."""
    if s.kind == 'P':
        return (int(s.batch_sizes.max()), int(s.batch_sizes.shape[0])) + tuple(s.data.shape[1:])
    hidden = s.data.shape[1:] if s.kind == 'C' else s.data.shape[2:]
    return (int(s.token_sizes.shape[0]), int(s.token_sizes.max())) + tuple(hidden)


def ptr(s: Seq):
    """cat.py:68-71, left.py:68-71, right.py:69-72, pack.py:23-27 -> (batch_ptr, token_ptr)."""
    if s.kind == 'P':
        batch_ptr, token_ptr = major_sizes_to_ptr(s.batch_sizes)
        return s.sorted_indices[batch_ptr], token_ptr
    token_ptr, batch_ptr = major_sizes_to_ptr(s.token_sizes)
    return batch_ptr, token_ptr


def raw(s: Seq):
    """cat.py:83, pack.py:51, left.py:83, right.py:85."""
    if s.kind in 'CP':
        return s.data
    return s.data.reshape((-1,) + s.data.shape[2:])


def idx(s: Seq) -> Seq:
    """cat.py:73-77, pack.py:33-37 (identity index in the same container);
    left.py:73-77, right.py:74-79 (flat padded index as a C)."""
    if s.kind in 'CP':
        return s.with_data(np.arange(s.data.shape[0], dtype=I64))
    t = size(s)[1]
    batch_ptr, token_ptr = ptr(s)
    flat = token_ptr + batch_ptr * t
    if s.kind == 'R':
        flat = flat + (t - s.token_sizes[batch_ptr])
    return C(flat, s.token_sizes)


def offsets(s: Seq):
    """cat.py:79-81, pack.py:43-45 (clamped to n-1); left.py:79-81, right.py:81-83 (b*t)."""
    if s.kind == 'C':
        return np.minimum(get_offsets(s.token_sizes), s.data.shape[0] - 1)
    if s.kind == 'P':
        return np.minimum(get_offsets(s.batch_sizes), s.data.shape[0] - 1)
    b, t = size(s)[:2]
    return np.arange(b, dtype=I64) * t


# --------------------------------------------------------------------------- core/view.py
def get_mask(s: Seq):
    """core/view.py:11-18."""
    b, t = size(s)[:2]
    batch_ptr, token_ptr = ptr(s)
    mask = np.zeros((b, t), I64)
    mask[batch_ptr, token_ptr] = 1
    return mask


def cat_view(s: Seq) -> Seq:
    """core/view.py:21-31."""
    if s.kind == 'C':
        return s
    return C(s.data, get_mask(s).sum(axis=1))


def _padded_view(s: Seq, kind: str, fill, dtype=None) -> Seq:
    """core/view.py:34-44 (left_view), 67-77 (right_view)."""
    if s.kind == kind:
        return s
    dtype = s.data.dtype if dtype is None else dtype
    return Seq(kind, _full(size(s), fill, dtype), token_sizes=get_mask(s).sum(axis=1))


def pack_view(s: Seq, sorted_indices) -> Seq:
    """core/view.py:47-64; `sorted_indices` = the reference's host torch.sort(descending=True)."""
    if s.kind == 'P':
        return s
    sorted_indices = _i64(sorted_indices)
    return P(s.data, get_mask(s).sum(axis=0), sorted_indices, invert_permutation(sorted_indices))


# --------------------------------------------------------------------------- core/get.py, set.py
def _flat_key(s: Seq, batch_ptr, token_ptr):
    """logical (batch_ptr, token_ptr) -> flat row of s's storage.
    C get.py:25-26, L get.py:41-42, P get.py:57-58, R get.py:73-74."""
    batch_ptr, token_ptr = _i64(batch_ptr), _i64(token_ptr)
    if s.kind == 'C':
        return offsets(s)[batch_ptr] + token_ptr
    if s.kind == 'P':
        return s.unsorted_indices[batch_ptr] + offsets(s)[token_ptr]
    t_phys = s.data.shape[1]
    if s.kind == 'L':
        return batch_ptr * t_phys + token_ptr
    return batch_ptr * t_phys + (size(s)[1] - s.token_sizes[batch_ptr] + token_ptr)


def getitem(s: Seq, key):
    """core/get.py:21-82: key is a Seq (re-wrap), a (batch_ptr, token_ptr) tuple, or a flat index."""
    if isinstance(key, Seq):
        return key.with_data(_gather(raw(s), key.data))
    if isinstance(key, tuple):
        return _gather(raw(s), _flat_key(s, *key))
    return _gather(raw(s), key)


def setitem(s: Seq, key, value) -> None:
    """core/set.py:21-92 (in place on s.data)."""
    flat = raw(s)
    assert np.shares_memory(flat, s.data), 'setitem needs contiguous storage'
    if isinstance(key, Seq):
        _scatter(flat, key.data, value)
    elif isinstance(key, tuple):
        _scatter(flat, _flat_key(s, *key), value)
    else:
        _scatter(flat, key, value)


# --------------------------------------------------------------------------- core/cast.py
def to_cat(s: Seq) -> Seq:
    """core/cast.py:8-16."""
    if s.kind == 'C':
        return s
    z = cat_view(s)
    return z.with_data(getitem(s, ptr(z)))


def to_pack(s: Seq, sorted_indices) -> Seq:
    """core/cast.py:41-49."""
    if s.kind == 'P':
        return s
    z = pack_view(s, sorted_indices)
    return z.with_data(getitem(s, ptr(z)))


def _to_padded(s: Seq, kind: str, fill=0) -> Seq:
    """core/cast.py:19-38 (left), 52-71 (right)."""
    if s.kind == kind:
        return s
    z = _padded_view(s, kind, fill)
    if s.kind in 'CP':       # cat_pack_to_left / cat_pack_to_right
        setitem(z, ptr(s), s.data)
    else:                    # right_to_left / left_to_right
        batch_ptr, token_ptr = ptr(s)
        setitem(z, (batch_ptr, token_ptr), getitem(s, (batch_ptr, token_ptr)))
    return z


def to_left(s: Seq, fill=0) -> Seq:
    return _to_padded(s, 'L', fill)


def to_right(s: Seq, fill=0) -> Seq:
    return _to_padded(s, 'R', fill)


def to_kind(s: Seq, kind: str, fill=0, sorted_indices=None) -> Seq:
    if kind == 'C':
        return to_cat(s)
    if kind == 'P':
        return to_pack(s, sorted_indices)
    return _to_padded(s, kind, fill)


# --------------------------------------------------------------------------- select/*.py
def last(s: Seq):
    """select/last.py:7-19."""
    b = size(s)[0]
    return getitem(s, (np.arange(b, dtype=I64), cat_view(s).token_sizes - 1))


def head(s: Seq, n: int) -> Seq:
    """select/head.py: C 6-19 (split/cat), P 22-33 (slice), L 36-45 (slice), R 48-67 (split/stack)."""
    if s.kind == 'C':
        off = get_offsets(s.token_sizes)
        rows = (off[:, None] + np.arange(n, dtype=I64)[None, :]).reshape(-1)
        return C(_gather(s.data, rows), np.full_like(s.token_sizes, n))
    if s.kind == 'P':
        return replace(s, data=s.data[:int(s.batch_sizes[0]) * n], batch_sizes=s.batch_sizes[:n])
    if s.kind == 'L':
        return L(s.data[:, :n], np.full_like(s.token_sizes, n))
    b, t = s.data.shape[:2]
    rows = (np.arange(b, dtype=I64) * t + (t - s.token_sizes))[:, None] + np.arange(n, dtype=I64)[None, :]
    data = _gather(raw(s), rows.reshape(-1)).reshape((b, n) + s.data.shape[2:])
    return R(data, np.full_like(s.token_sizes, n))


def roll(s: Seq, shifts: int, sorted_indices=None) -> Seq:
    """select/roll.py: C 6-16; L 19-23, P 26-30, R 33-37 = self[self.idx().cat().roll(s).<layout>()]."""
    if s.kind == 'C':
        batch_ptr, token_ptr = ptr(s)
        sizes = np.repeat(s.token_sizes, s.token_sizes)
        token_ptr = np.mod(token_ptr - shifts + sizes, sizes)
        return s.with_data(getitem(s, (batch_ptr, token_ptr)))
    index = roll(to_cat(idx(s)), shifts)
    if s.kind == 'P':
        sorted_indices = s.sorted_indices if sorted_indices is None else sorted_indices
    return getitem(s, to_kind(index, s.kind, 0, sorted_indices))


def rev(s: Seq, sorted_indices=None) -> Seq:
    """select/rev.py: C 6-22, L 25-29, P 32-36, R 39-41 (all equal a per-sequence flip)."""
    if s.kind == 'C':
        batch_ptr, token_ptr = ptr(s)
        sizes = np.repeat(s.token_sizes, s.token_sizes)
        return s.with_data(getitem(s, (batch_ptr, sizes - 1 - token_ptr)))
    if s.kind == 'L':
        return to_left(R(np.ascontiguousarray(s.data[:, ::-1]), s.token_sizes))
    if s.kind == 'R':
        return to_right(L(np.ascontiguousarray(s.data[:, ::-1]), s.token_sizes))
    index = rev(to_cat(idx(s)))
    return getitem(s, to_pack(index, s.sorted_indices if sorted_indices is None else sorted_indices))


def trunc(s: Seq, ab) -> Seq:
    """select/trunc.py: C 9-22, L 25-35, P 38-47, R 50-62."""
    a, b = ab
    if s.kind == 'C':
        off = get_offsets(s.token_sizes)
        new = s.token_sizes - a - b
        tok, bat = major_sizes_to_ptr(new)
        return C(_gather(s.data, off[bat] + a + tok), new)
    if s.kind == 'P':
        bs = s.batch_sizes[a + b:]
        batch_ptr, token_ptr = major_sizes_to_ptr(bs)
        return replace(s, data=_gather(s.data, batch_ptr + offsets(s)[token_ptr + a]), batch_sizes=bs)
    t = size(s)[1]
    return Seq(s.kind, s.data[:, a:t - b], token_sizes=s.token_sizes - a - b)


def mask(s: Seq, zero, one, dtype):
    """mask.py:6-14."""
    b, t = size(s)[:2]
    m = np.full((b, t), zero, dtype)
    m[ptr(s)] = one
    return m


# --------------------------------------------------------------------------- reduce.py
_OPS = {'sum': 0, 'mean': 1, 'max': 2, 'min': 3, 'prod': 4}


def _segment_reduce(data, lens, op: str, initial):
    """torch.segment_reduce(data, op, lengths=lens, unsafe=True, initial=initial) — reduce.py:36-53."""
    data = np.ascontiguousarray(data)
    assert data.dtype in (np.float32, np.float64), 'oracle reduces in f32/f64 (SURVEY §8c)'
    lens = _i64(lens)
    S = lens.size
    H = int(np.prod(data.shape[1:], dtype=np.int64))
    out = np.empty((S,) + data.shape[1:], data.dtype)
    off = get_offsets(lens)
    if data.dtype == np.float32:
        lib().orc_segment_reduce_f32(_p(data), _p(lens), _p(off), ctypes.c_int64(S), ctypes.c_int64(H),
                                     ctypes.c_int(_OPS[op]), ctypes.c_float(initial), _p(out))
    else:
        lib().orc_segment_reduce_f64(_p(data), _p(lens), _p(off), ctypes.c_int64(S), ctypes.c_int64(H),
                                     ctypes.c_int(_OPS[op]), ctypes.c_double(initial), _p(out))
    return out


def _global(fn, data):
    """tensor.min()/max() as used for `initial` (reduce.py:35,40); torch propagates NaN."""
    if data.size == 0:
        raise ValueError('min/max of an empty tensor (the reference raises here too)')
    return float('nan') if np.isnan(data).any() else float(fn(data))


def segment_max(data, lens):
    """reduce.py:34-36."""
    return _segment_reduce(data, lens, 'max', _global(np.min, data))


def segment_min(data, lens):
    """reduce.py:39-41."""
    return _segment_reduce(data, lens, 'min', _global(np.max, data))


def segment_sum(data, lens):
    """reduce.py:44-45."""
    return _segment_reduce(data, lens, 'sum', 0.0)


def segment_mean(data, lens):
    """reduce.py:48-49."""
    return _segment_reduce(data, lens, 'mean', 0.0)


def segment_prod(data, lens):
    """reduce.py:52-53."""
    return _segment_reduce(data, lens, 'prod', 1.0)


def segment_logsumexp(data, lens):
    """reduce.py:56-61."""
    lens = _i64(lens)
    m = segment_max(data, lens)
    with np.errstate(all='ignore'):
        shifted = np.exp(data - np.repeat(m, lens, axis=0))
        eps = (lens == 0).astype(data.dtype).reshape((-1,) + (1,) * (data.ndim - 1))
        return np.log(segment_sum(shifted.astype(data.dtype), lens) + eps) + m


def segment_head(data, lens):
    """reduce.py:64-65."""
    return head(C(data, lens), 1).data


def segment_last(data, lens):
    """reduce.py:68-69."""
    return last(C(data, lens))


_INT_KINDS = {np.dtype(np.int64): 0, np.dtype(np.int32): 1, np.dtype(np.int16): 2, np.dtype(np.int8): 3,
              np.dtype(np.uint8): 4}


def _index_reduce_int(tensor, index, source, op: int, include_self: bool):
    """reduce.py:6-23 on integer tensors: ATen's own steps in the tensor's integer type (rua_oracle.c)."""
    tensor = np.array(tensor, order='C', copy=True)
    kind = _INT_KINDS[tensor.dtype]
    source = np.ascontiguousarray(source, dtype=tensor.dtype)
    index = _i64(index)
    S = tensor.shape[0]
    H = int(np.prod(tensor.shape[1:], dtype=np.int64))
    assert index.size == 0 or (index.min() >= 0 and index.max() < S), 'index out of range'
    counts = np.empty(S, I64)
    lib().orc_index_reduce_int(_p(tensor), ctypes.c_int64(S), ctypes.c_int64(H), _p(index), _p(source),
                               ctypes.c_int64(index.size), ctypes.c_int(op), ctypes.c_int(int(include_self)),
                               ctypes.c_int(kind), _p(counts))
    return tensor


def _index_reduce(tensor, index, source, op: int, include_self: bool):
    if np.asarray(tensor).dtype in _INT_KINDS:
        return _index_reduce_int(tensor, index, source, op, include_self)
    tensor = np.array(tensor, dtype=np.float32, order='C', copy=True)
    source = np.ascontiguousarray(source, dtype=np.float32)
    index = _i64(index)
    S = tensor.shape[0]
    H = int(np.prod(tensor.shape[1:], dtype=np.int64))
    counts = np.empty(S, I64)
    lib().orc_index_reduce_f32(_p(tensor), ctypes.c_int64(S), ctypes.c_int64(H), _p(index), _p(source),
                               ctypes.c_int64(index.size), ctypes.c_int(op), ctypes.c_int(int(include_self)),
                               _p(counts))
    return tensor


def scatter_max(tensor, index, source, include_self=False):
    """reduce.py:6-7."""
    return _index_reduce(tensor, index, source, 2, include_self)


def scatter_min(tensor, index, source, include_self=False):
    """reduce.py:10-11."""
    return _index_reduce(tensor, index, source, 3, include_self)


def scatter_sum(tensor, index, source, include_self=False):
    """reduce.py:14-15 (index_add into tensor, or into zeros)."""
    t = np.asarray(tensor)
    base = tensor if include_self else np.zeros_like(t if t.dtype in _INT_KINDS else np.asarray(tensor, dtype=np.float32))
    return _index_reduce(base, index, source, 0, True)


def scatter_mean(tensor, index, source, include_self=False):
    """reduce.py:18-19."""
    return _index_reduce(tensor, index, source, 1, include_self)


def scatter_prod(tensor, index, source, include_self=False):
    """reduce.py:22-23."""
    return _index_reduce(tensor, index, source, 4, include_self)


def scatter_logsumexp(tensor, index, source, include_self=False):
    """reduce.py:26-31.  Integer tensors: the maximum and both differences are taken IN THE INTEGER TYPE (they wrap
    there, as torch's do), `.exp()` promotes to float32, and so does `+ m` at the end."""
    if np.asarray(tensor).dtype in _INT_KINDS:
        tensor, source = np.asarray(tensor), np.asarray(source, dtype=np.asarray(tensor).dtype)
        m = scatter_max(tensor, index, source, include_self)
        with np.errstate(all='ignore'):
            t = np.exp((tensor - m).astype(np.float32))              # (numpy integer subtraction wraps like torch's)
            s = np.exp((source - m[_i64(index)]).astype(np.float32))
            return np.log(scatter_sum(t, index, s, include_self)) + m.astype(np.float32)
    tensor = np.asarray(tensor, dtype=np.float32)
    source = np.asarray(source, dtype=np.float32)
    m = scatter_max(tensor, index, source, include_self)
    with np.errstate(all='ignore'):
        t = np.exp(tensor - m)
        s = np.exp(source - m[_i64(index)])
        return np.log(scatter_sum(t, index, s, include_self)) + m


# --------------------------------------------------------------------------- segment.py
def seg(s: Seq, duration: Seq, fn, sorted_indices=None, duration_sorted=None) -> Seq:
    """segment.py: C 6-13, L 16-28, P 31-35, R 38-50.  `fn` is one of the segment_* above."""
    if s.kind == 'C':
        d = to_cat(duration)
        return d.with_data(fn(s.data, d.data))
    if s.kind == 'P':
        out = seg(to_cat(s), duration, fn)
        return to_pack(out, sorted_indices)
    b, t = size(s)[:2]
    hidden = size(s)[2:]
    if s.kind == 'L':
        d = to_left(duration, 0)
        sizes = np.concatenate([d.data, (t - s.token_sizes)[:, None]], axis=-1).reshape(-1)
    else:
        d = to_right(duration, 0)
        sizes = np.concatenate([(t - s.token_sizes)[:, None], d.data], axis=-1).reshape(-1)
    data = fn(np.ascontiguousarray(raw(s)), sizes).reshape((b, -1) + tuple(hidden))
    data = data[:, :-1] if s.kind == 'L' else data[:, 1:]
    return Seq(s.kind, data, token_sizes=d.token_sizes)


# --------------------------------------------------------------------------- core/__init__.py
def new(kind: str, arrays, fill=0, sorted_indices=None) -> Seq:
    """core/__init__.py:9-36: C.new = cat + sizes; L/P/R.new = C.new(...).<cast>()."""
    data = np.concatenate(arrays, axis=0)
    c = C(data, [a.shape[0] for a in arrays])
    return to_kind(c, kind, fill, sorted_indices)


# --------------------------------------------------------------------------- compose.py, detach.py
def compose(seqs, sort_desc) -> Seq:
    """compose.py:9-33, step by step.  `sort_desc(lens)` = the reference's host call
    torch.sort(lens, descending=True)[1] (core/view.py:48), which both of compose's pack() calls make."""
    offset, data, indices, token_sizes = 0, [], [], []
    for s in seqs:                                             # compose.py:12-19
        data.append(raw(s))
        flat = to_cat(idx(s))                                  # sequence.idx().cat()
        indices.append(flat.data + offset)
        token_sizes.append(flat.token_sizes)
        offset += data[-1].shape[0]
    ts = new('C', token_sizes)                                 # compose.py:21  C.new(token_sizes)
    order = to_pack(idx(ts), sort_desc(ts.token_sizes)).data   # compose.py:22  unsorted_indices, _, _, _ = ....idx().pack()
    ind = to_pack(C(np.concatenate(indices), ts.data), sort_desc(ts.data))          # compose.py:24
    unsorted = ind.unsorted_indices[order]                     # compose.py:25
    ind = replace(ind, sorted_indices=invert_permutation(unsorted), unsorted_indices=unsorted)   # compose.py:26-31
    return ind.with_data(_gather(np.concatenate(data, axis=0), ind.data))           # compose.py:33  tensor[Z], get.py:11-13


def split(s: Seq):
    """detach.py: C/P 9-17 (split of the cat form), L 22-28, R 33-40 (split of the flattened padded storage into
    (tokens, padding) pairs; raises, like torch.split, when the storage is wider than the longest sequence)."""
    if s.kind in 'CP':
        c = to_cat(s)
        return np.split(c.data, np.cumsum(c.token_sizes)[:-1])
    t = size(s)[1]
    pair = [s.token_sizes, t - s.token_sizes] if s.kind == 'L' else [t - s.token_sizes, s.token_sizes]
    sizes = np.stack(pair, axis=-1).reshape(-1)
    flat = raw(s)
    if int(sizes.sum()) != flat.shape[0]:
        raise RuntimeError(f'split_with_sizes expects split_sizes to sum exactly to {flat.shape[0]}')
    parts = np.split(flat, np.cumsum(sizes)[:-1])
    return parts[0::2] if s.kind == 'L' else parts[1::2]


def stable_descending_order(lens):
    """A valid (not necessarily the reference's) sorted_indices: descending, ties by index."""
    lens = _i64(lens)
    return np.argsort(-lens, kind='stable').astype(I64)
