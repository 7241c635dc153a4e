"""Views, indexing, casts and constructors — mirror of torchrua.core (reference core/view.py,
core/get.py, core/set.py, core/cast.py, core/__init__.py).

Every conversion between the four layouts is ONE launch of the row mover (rua_move_rows): the
destination rows are enumerated in storage order, each row finds its source row in closed form, and
padding is written in the same pass (the reference pre-fills with new_full and then scatters:
core/view.py:34-38 + core/cast.py:19-23).
"""
import threading
from numbers import Number
from typing import Any, List, Tuple, Union

import torch
from torch import Tensor

from torchrua_amd import _lib as K
from torchrua_amd import _meta as M
from torchrua_amd import _ops as O
from torchrua_amd.layout import C, L, P, R, T, Z, describe, lens_of

__all__ = ['get_mask']
# NOTE: _namespace.py gives this module the attributes `cast`, `get`, `set`, `view` (the reference's submodules of
# torchrua.core) — which shadow the builtin `set` here, as they do in the reference's own core/__init__.py.  Spell a set
# literal {..} / builtins.set if this file ever needs one.

Key = Union[int, Tensor, Tuple[Tensor, Tensor], Z]


def _to_self(self: Any, *_, **__) -> Any:
    return self


# ------------------------------------------------------------------ core/view.py
def get_mask(self: Z) -> Tensor:
    """core/view.py:11-18: [b, t] int64 grid of 0/1 — one kernel, no (batch_ptr, token_ptr) scatter."""
    lens = lens_of(self)
    # (rows = every sequence of the batch: P.size() counts only the non-empty ones, layout/pack.py:12-17)
    return _mask_grid(lens, lens.numel(), self.size()[1], 0, 1, torch.long)


def _mask_grid(lens: Tensor, b: int, t: int, zero, one, dtype: torch.dtype) -> Tensor:
    dev = K.require_device(lens)
    lib = K.load()
    out = torch.empty((b, t), dtype=dtype, device=dev)
    es = out.element_size()
    if es not in (1, 2, 4, 8):
        raise K.RuaError(f'mask dtype {dtype} not supported')
    zb = int.from_bytes(O._fill16(zero, dtype)[:es], 'little')
    ob = int.from_bytes(O._fill16(one, dtype)[:es], 'little')
    K.check(lib.rua_mask(K.ptr(M._as_lens(lens)), b, t, K.ptr(out), es, zb, ob, K.stream_ptr(dev)), 'rua_mask')
    return out


def _cat_view(self: Union[L, P, R], **kwargs) -> C:
    """core/view.py:21-25."""
    return C(data=self.data, token_sizes=lens_of(self))


def _padded_view(cls):
    def view(self, fill_value: Number, dtype: torch.dtype = None):
        """core/view.py:34-38 / 67-71."""
        data = self.data.new_full(self.size(), fill_value=fill_value, dtype=dtype)
        return cls(data=data, token_sizes=lens_of(self))
    return view


def _pack_meta(token_sizes: Tensor, dev: torch.device):
    """sorted/unsorted indices (device), batch_sizes (CPU, as PackedSequence mandates), and the
    device-side boff — K3 + K1.  One host sort, no B x T mask (core/view.py:47-58)."""
    lib = K.load()
    lens = M._as_lens(token_sizes)
    # the same lengths packed again (a container reused across steps, or inside a HIP-graph capture): no host
    # sort, no upload — unless somebody wrote into the tensors handed out last time
    key = f'pack_meta:{dev}'
    hit = M._memo_get(token_sizes, key)
    if hit is not None and all(M._version(t) == v for t, v in zip(hit[0], hit[1])):
        return (lens,) + hit[0]
    # The reference's order: a HOST sort of the lengths, descending (core/view.py:48).  Its tie order is
    # implementation-defined, so bit-exact parity means reproducing that very call (SURVEY.md §8a note;
    # _meta.host_sort_desc).  With device-only lengths the GPU idles from the read-back until the mover below is
    # launched, so nothing else sits between the two.
    read_back = token_sizes.is_cuda and M._memo_get(token_sizes, 'host') is None      # device-only lengths: a sync now
    host = M.host_lens(token_sizes)
    B = lens.numel()
    if read_back and B >= 4096 and M.host_sort_is_native():
        meta = _pack_meta_overlapped(token_sizes, lens, host, dev)
        M._memo_put(token_sizes, key, (meta, tuple(M._version(t) for t in meta)))
        M._memo_put(meta[0], 'reference_order', True)
        return (lens,) + meta
    sorted_indices = M.sorted_indices_to_device(host, dev, stream_idle=read_back)
    T = M.max_len(token_sizes)
    batch_sizes = M.batch_sizes_from_host_lens(host, T)
    # one call for everything derived on the device; the internal vectors share one allocation:
    # batch_sizes [T] | their offsets [T] | offsets of the lengths [B] | scan scratch
    need_off = M._memo_get(token_sizes, 'off') is None and lens is token_sizes
    unsorted = torch.empty(B, dtype=torch.long, device=dev)
    buf = torch.empty(2 * T + B + lib.rua_scan_ws_elems(max(B, T)), dtype=torch.long, device=dev)
    bsz_dev, boff, off = buf[:T], buf[T:2 * T], buf[2 * T:2 * T + B]
    K.check(lib.rua_pack_prepare(K.ptr(lens), K.ptr(sorted_indices), B, T, K.ptr(unsorted), K.ptr(bsz_dev),
                                 K.ptr(boff), K.ptr(off) if need_off else None, K.ptr(buf[2 * T + B:]),
                                 K.stream_ptr(dev)), 'rua_pack_prepare')
    if need_off:
        M._memo_put(token_sizes, 'off', off)      # the CattedSequence side of the cast needs them next
    # (`lens` itself stays out of the memo: a tensor whose memo holds the tensor is a reference cycle, and cyclic
    # garbage made at every step drives Python into full collections — 38 ms each with torch's heap to scan)
    meta = (sorted_indices, unsorted, batch_sizes, bsz_dev, boff)
    M._memo_put(token_sizes, key, (meta, tuple(M._version(t) for t in meta)))
    M._memo_put(sorted_indices, 'reference_order', True)        # (pack_reference_order below)
    return (lens,) + meta


_sort_job = threading.Lock()


def _pack_meta_overlapped(token_sizes: Tensor, lens: Tensor, host: Tensor, dev: torch.device):
    """_pack_meta for DEVICE-ONLY lengths (the reference's own constructor signature: C(data, token_sizes_on_device)).
    The read-back has just synchronised the stream and the GPU idles until the mover is launched, so every host
    microsecond from here to that launch shows (profiles/r03_devlens_probe.txt: 0.41 ms per pack(), the sort 0.22 of
    them).  The sort therefore runs on the library's helper thread (rua_host_sort_desc_begin) WHILE this thread does the
    rest — batch_sizes, both offset scans (on the host here: the order is not needed for them, so the device-side
    rua_pack_prepare launch disappears from the critical path), the allocations — and everything goes up in two async
    copies; only the inverse permutation is left to a (tiny) launch."""
    lib = K.load()
    B = lens.numel()
    host = M._as_lens(host.detach())          # (int32 / strided lengths: the C side reads contiguous int64)
    staged_order = torch.empty(B, dtype=torch.long, pin_memory=True)
    _sort_job.acquire()            # the helper takes one job at a time (ctypes drops the GIL: another host thread may be here)
    try:
        K.check(lib.rua_host_sort_desc_begin(host.data_ptr(), B, staged_order.data_ptr(), M.host_sort_threads()),
                'rua_host_sort_desc_begin')
    except BaseException:
        _sort_job.release()
        raise
    try:
        T = M.max_len(token_sizes)
        batch_sizes = M.batch_sizes_from_host_lens(host, T)
        need_off = M._memo_get(token_sizes, 'off') is None and lens is token_sizes
        staged = torch.empty(2 * T + B, dtype=torch.long, pin_memory=True)      # batch_sizes | their offsets | offsets of the lengths
        staged[:T].copy_(batch_sizes)
        K.check(lib.rua_host_pack_scans(host.data_ptr(), B, batch_sizes.data_ptr(), T, staged.data_ptr() + 8 * T,
                                        staged.data_ptr() + 16 * T if need_off else None), 'rua_host_pack_scans')
        unsorted = torch.empty(B, dtype=torch.long, device=dev)
        sorted_indices = torch.empty(B, dtype=torch.long, device=dev)
        buf = torch.empty(2 * T + B, dtype=torch.long, device=dev)
    finally:
        rc = lib.rua_host_sort_desc_end()          # (always: the helper must be idle again whatever happened above)
        _sort_job.release()
    K.check(rc, 'rua_host_sort_desc_end')
    sorted_indices.copy_(staged_order, non_blocking=True)
    if need_off:
        buf.copy_(staged, non_blocking=True)
    else:
        buf[:2 * T].copy_(staged[:2 * T], non_blocking=True)
    bsz_dev, boff, off = buf[:T], buf[T:2 * T], buf[2 * T:]
    K.check(lib.rua_pack_meta(None, K.ptr(sorted_indices), B, 0, K.ptr(unsorted), None, K.stream_ptr(dev)), 'rua_pack_meta')
    if need_off:
        M._memo_put(token_sizes, 'off', off)
    return (sorted_indices, unsorted, batch_sizes, bsz_dev, boff)


def pack_reference_order(p: P):
    """The metadata `P.roll` / `P.rev` return in the reference: both end in `.pack()` (select/roll.py:26-30,
    select/rev.py:33-34), i.e. in a fresh host sort of the lengths (core/view.py:48), whatever order `p` itself is in.
    None when p's own order already is that one — always for a PackedSequence made by this library, and for
    torch's pack_sequence(enforce_sorted=False), which makes the same call; a hand-made PackedSequence whose ties sit
    in another order (a stable sort, say) is checked ONCE (a read-back of its lengths, as the reference pays on every
    call) and, if it differs, gets (lens, sorted, unsorted, batch_sizes, bsz_dev, boff) in the reference's order."""
    given = p.sorted_indices
    if given is None or M._memo_get(given, 'reference_order'):
        return None
    dev = K.require_device(p.data)
    meta = _pack_meta(M.pack_lens(p), dev)
    if torch.equal(meta[1], given):
        M._memo_put(given, 'reference_order', True)
        return None
    return meta


def _pack_view(self: Union[C, L, R], **kwargs) -> P:
    """core/view.py:47-58."""
    dev = K.require_device(self.data)
    lens, sorted_indices, unsorted, batch_sizes, bsz_dev, boff = _pack_meta(self.token_sizes, dev)
    p = P(data=self.data, batch_sizes=batch_sizes, sorted_indices=sorted_indices, unsorted_indices=unsorted)
    M.adopt_pack(p, lens, boff, bsz_dev)
    return p


C.cat_view = _to_self
L.cat_view = _cat_view
P.cat_view = _cat_view
R.cat_view = _cat_view

C.left_view = _padded_view(L)
L.left_view = _to_self
P.left_view = _padded_view(L)
R.left_view = _padded_view(L)

C.pack_view = _pack_view
L.pack_view = _pack_view
P.pack_view = _to_self
R.pack_view = _pack_view

C.right_view = _padded_view(R)
L.right_view = _padded_view(R)
P.right_view = _padded_view(R)
R.right_view = _to_self


# ------------------------------------------------------------------ core/cast.py
def _hidden(z: Z) -> Tuple[int, ...]:
    return tuple(z.data.shape[1:]) if isinstance(z, (C, P)) else tuple(z.data.shape[2:])


def _to_cat(self: Union[L, P, R]) -> C:
    """core/cast.py:8-16: L/P/R -> C."""
    lens = lens_of(self)
    B = lens.numel()
    n = int(self.data.size(0)) if isinstance(self, P) else M.total_len(lens)
    dst = M.lay_cat(lens, B, n)
    data = O.move(self.data, O.MovePlan(dst, describe(self), (n,) + _hidden(self), name='to_cat'))
    return C(data=data, token_sizes=lens)


def _to_padded(cls, kind):
    def cast(self, fill_value: Number = 0):
        """core/cast.py:19-38 (left) / 52-71 (right): fill and payload in one pass."""
        lens = lens_of(self)
        b, t = lens.numel(), self.size()[1]
        dst = M.lay_padded(kind, lens, b, t, t)
        src = describe(self)
        if isinstance(self, P) and 0 < M.row_bytes(self.data, 1) <= M.NARROW_ROW_BYTES:
            # narrow rows out of a PackedSequence: (rank x time) tiles over the destination's whole grid (tokens + fill)
            src = M.lay_pack(self, row_bytes=M.row_bytes(self.data, 1), full_grid_T=t)
        plan = O.MovePlan(dst, src, (b, t) + _hidden(self), fill=fill_value,
                          name='to_left' if kind == K.LEFT else 'to_right')
        return cls(data=O.move(self.data, plan), token_sizes=lens)
    return cast


def _to_pack(self: Union[C, L, R]) -> P:
    """core/cast.py:41-49: C/L/R -> P."""
    dev = K.require_device(self.data)
    lens, sorted_indices, unsorted, batch_sizes, bsz_dev, boff = _pack_meta(self.token_sizes, dev)
    n = int(self.data.size(0)) if isinstance(self, C) else M.total_len(self.token_sizes)
    shell = P(data=self.data, batch_sizes=batch_sizes, sorted_indices=sorted_indices, unsorted_indices=unsorted)
    M.adopt_pack(shell, lens, boff, bsz_dev)
    dst = M.lay_pack(shell, lens=lens, boff=boff, T=batch_sizes.numel(), n_rows=n,
                     row_bytes=M.row_bytes(self.data, 1 if isinstance(self, C) else 2))
    data = O.move(self.data, O.MovePlan(dst, describe(self), (n,) + _hidden(self), name='to_pack'))
    return shell._replace(data=data)


C.cat = _to_self
L.cat = _to_cat
P.cat = _to_cat
R.cat = _to_cat

C.left = _to_padded(L, K.LEFT)
L.left = _to_self
P.left = _to_padded(L, K.LEFT)
R.left = _to_padded(L, K.LEFT)

C.pack = _to_pack
L.pack = _to_pack
P.pack = _to_self
R.pack = _to_pack

C.right = _to_padded(R, K.RIGHT)
L.right = _to_padded(R, K.RIGHT)
P.right = _to_padded(R, K.RIGHT)
R.right = _to_self


# ------------------------------------------------------------------ core/get.py / core/set.py
_tensor_getitem = Tensor.__getitem__       # torch's own, whatever patch_tensor_indexing() does later
_tensor_setitem = Tensor.__setitem__


def _is_ptr_pair(key) -> bool:
    return isinstance(key, tuple) and len(key) == 2 and isinstance(key[0], Tensor) and isinstance(key[1], Tensor)


def _flat_rows(z: Z, key: Tuple[Tensor, Tensor]) -> Tensor:
    """Flat storage rows of (batch_ptr, token_ptr) — core/get.py:25-26, 41-42, 57-58, 73-74 —
    produced by the mover itself (moving an iota of the storage)."""
    n_rows = describe(z).n_rows
    iota = _iota(n_rows, z.data.device)
    return O.launch_move(O.MovePlan(M.lay_list(*key), describe(z), (key[0].numel(),), name='flat_rows'), iota)


def _iota(n: int, dev: torch.device) -> Tensor:
    """0 .. n - 1 as ONE launch of K2: the flat rows of a single sequence of n tokens (round 4 scanned n ones: two
    n-sized tensors and a three-pass scan per autograd `X[batch_ptr, token_ptr]`)."""
    out = torch.empty(n, dtype=torch.long, device=dev)
    if n:
        K.check(K.load().rua_enum_rows(M.lay_cat(None, 1, n, len_add=n).ref(), n, None, None, K.ptr(out),
                                       K.stream_ptr(dev)), 'rua_enum_rows')
    return out


def _row_index(index: Tensor, n_rows: int):
    """An index tensor as int64 row numbers, or None when it is not a plain row index.  Narrower integer types are
    widened (an index-sized cast; the payload still moves through the row mover); a 1-D bool / uint8 mask over the
    rows selects the rows where it is set, like torch (the count is data-dependent: one sync, as in torch)."""
    if index.dtype == torch.long:
        return index
    if index.dtype in (torch.int32, torch.int16, torch.int8):
        return index.to(torch.long)
    if index.dtype in (torch.bool, torch.uint8) and index.dim() == 1 and index.numel() == n_rows:
        return torch.nonzero(index).reshape(-1)
    return None


def _gather_flat(raw: Tensor, index: Tensor) -> Tensor:
    """raw[index] for a row index of any shape (core/get.py:29,42,61,74) via the mover."""
    rows = _row_index(index, int(raw.size(0)))
    if rows is None:
        return _tensor_getitem(raw, index)   # multi-dimensional masks etc.: not a row-index gather
    index = rows
    flat = index.reshape(-1)
    shape = tuple(index.shape) + tuple(raw.shape[1:])
    plan = O.MovePlan(M.lay_list(None, flat), M.lay_flat(int(raw.size(0))), shape, name='gather_flat')
    if raw.requires_grad and torch.is_grad_enabled():
        return O._ListGather.apply(raw, plan, lambda: flat)
    return O.launch_move(plan, raw)


def _needs_autograd(raw: Tensor, value) -> bool:
    """Would torch record this in-place write?  (the mover writes into raw.detach(); a gradient to `value`
    or through a non-leaf `raw` would be dropped, a leaf that requires grad would be overwritten without torch's
    error.)  Such writes take torch's own setitem — what the reference does (core/set.py:10-18)."""
    return torch.is_grad_enabled() and (raw.requires_grad or (isinstance(value, Tensor) and value.requires_grad))


def _written(raw: Tensor) -> None:
    """The mover wrote into raw's storage behind autograd's back: bump the version counter, so that autograd's
    saved-tensor check and this library's own version-keyed memos see the write."""
    if not raw.is_inference():
        torch.autograd.graph.increment_version(raw)


def _scatter_flat(raw: Tensor, index: Tensor, value) -> None:
    """raw[index] = value (core/set.py:30,45,67,82) via the mover in scatter mode."""
    rows = _row_index(index, int(raw.size(0)))
    if rows is None or not raw.is_contiguous() or _needs_autograd(raw, value):
        _tensor_setitem(raw, index, value)
        return
    index = rows
    flat = index.reshape(-1)
    value = torch.as_tensor(value, dtype=raw.dtype, device=raw.device)
    value = value.detach().expand(tuple(index.shape) + tuple(raw.shape[1:])).contiguous()
    plan = O.MovePlan(M.lay_list(None, flat), M.lay_flat(int(raw.size(0))), raw.shape, flags=K.MOVE_SCATTER,
                      name='scatter_flat')
    O.launch_move(plan, value, out=raw.detach())
    _written(raw)


def _getitem(cls):
    base = tuple.__getitem__

    def getitem(self, key: Key):
        if isinstance(key, (C, L, P, R)):          # Z key: gather rows by a container of row indices
            return key._replace(data=_gather_flat(self.raw(), key.data))
        if _is_ptr_pair(key):                       # (batch_ptr, token_ptr)
            bp, tp = key
            shape = tuple(bp.shape) + _hidden(self)
            plan = O.MovePlan(M.lay_list(bp.reshape(-1), tp.reshape(-1)), describe(self), shape, name='getitem')
            if self.data.requires_grad and torch.is_grad_enabled():
                k = (bp.reshape(-1), tp.reshape(-1))
                return O._ListGather.apply(self.data, plan, lambda: _flat_rows(self, k), len(self.data.shape) - len(_hidden(self)))
            return O.launch_move(plan, self.data)
        if isinstance(key, Tensor):
            return _gather_flat(self.raw(), key)
        return base(self, key)
    return getitem


def _setitem(cls):
    def setitem(self, key: Key, value: Tensor) -> None:
        if isinstance(key, (C, L, P, R)):
            _scatter_flat(self.raw(), key.data, value)
            return None
        if _is_ptr_pair(key):
            bp, tp = key[0].reshape(-1), key[1].reshape(-1)
            if not self.data.is_contiguous():
                raise K.RuaError('__setitem__ needs contiguous storage')
            hidden = _hidden(self)
            if _needs_autograd(self.data, value):
                # autograd has to see the write: torch's own setitem on the flat rows the kernels compute
                _tensor_setitem(self.raw(), _flat_rows(self, (bp, tp)), value)
                return None
            value = torch.as_tensor(value, dtype=self.data.dtype, device=self.data.device)
            value = value.detach().expand((bp.numel(),) + hidden).contiguous()
            plan = O.MovePlan(M.lay_list(bp, tp), describe(self), self.data.shape, flags=K.MOVE_SCATTER,
                              name='setitem')
            O.launch_move(plan, value, out=self.data.detach())
            _written(self.data)
            return None
        if isinstance(key, Tensor):
            _scatter_flat(self.raw(), key, value)
            return None
        raise TypeError(f'{cls.__name__} does not support item assignment with key {type(key).__name__}')
    return setitem


for _cls in (C, L, P, R):
    _cls.__getitem__ = _getitem(_cls)
    _cls.__setitem__ = _setitem(_cls)


def tensor_getitem(self: T, key):
    """core/get.py:11-18: `tensor[Z]` re-wraps a container of row indices; any other key is torch's own indexing.
    Payload on a HIP device goes through the row mover, like container[Z]."""
    if isinstance(key, (C, L, P, R)):
        if self.is_cuda and self.dim() >= 1:
            return key._replace(data=_gather_flat(self, key.data))
        return key._replace(data=_tensor_getitem(self, key.data))
    return _tensor_getitem(self, key)


def tensor_setitem(self: T, key, value) -> None:
    """core/set.py:10-18: `tensor[Z] = value` scatters through a container of row indices."""
    if isinstance(key, (C, L, P, R)):
        if self.is_cuda and self.dim() >= 1:
            return _scatter_flat(self, key.data, value)
        return _tensor_setitem(self, key.data, value)
    return _tensor_setitem(self, key, value)


def patch_tensor_indexing() -> None:
    """Twin of the reference's import-time patch of Tensor.__getitem__/__setitem__ (core/get.py:11-18,
    core/set.py:10-18): lets `tensor[Z]` re-wrap a container of row indices and `tensor[Z] = value` scatter through
    one.  `install_as_torchrua()` applies it (importing the reference does); a plain `import torchrua_amd` does not,
    so that ordinary tensor indexing in the process is left untouched."""
    Tensor.__getitem__ = tensor_getitem
    Tensor.__setitem__ = tensor_setitem


def unpatch_tensor_indexing() -> None:
    """Put torch's own Tensor.__getitem__/__setitem__ back."""
    Tensor.__getitem__ = _tensor_getitem
    Tensor.__setitem__ = _tensor_setitem


# ------------------------------------------------------------------ core/__init__.py constructors
def _new_cat(tensors: List[T]) -> C:
    """core/__init__.py:9-15.  The lengths are known on the host here: keep that copy as the
    host mirror so that later pack()/left()/size() never read them back from the device."""
    data = torch.cat(tensors, dim=0)
    host = torch.tensor([tensor.size()[0] for tensor in tensors], dtype=torch.long)
    token_sizes = M.to_device_async(host, data.device)
    M.attach_host(token_sizes, host)
    return C(data=data, token_sizes=token_sizes)


C.new = staticmethod(_new_cat)
L.new = staticmethod(lambda tensors, fill_value=0: _new_cat(tensors).left(fill_value=fill_value))
P.new = staticmethod(lambda tensors: _new_cat(tensors).pack())
R.new = staticmethod(lambda tensors, fill_value=0: _new_cat(tensors).right(fill_value=fill_value))


def with_host_sizes(data: Tensor, token_sizes_host: Tensor) -> C:
    """C(data, token_sizes) from lengths that live on the host (what C.new does for a list).  The library keeps its
    OWN pinned copy as the host mirror — the caller may refill its buffer at once, by any means — and uploads from that
    copy, so the privacy costs no extra pass (see _meta.private_host_copy)."""
    K.require_device(data)
    mirror = M.private_host_copy(token_sizes_host)
    dev_sizes = M.pinned_to_device_async(mirror, data.device)
    M.attach_host(dev_sizes, mirror)
    return C(data=data, token_sizes=dev_sizes)
