"""Host-side bookkeeping for the kernels: derived index vectors (offsets, batch offsets, lengths of
a PackedSequence), host mirrors of length vectors, and `rua_layout` descriptors.

Derived vectors are produced by the HIP kernels (K1/K3/K3b in include/rua.h) and memoised ON THE
TENSOR OBJECT they were derived from (guarded by the tensor's `_version`), so a chain like
`c.pack().roll(1).cat()` — forward and backward — scans each length vector once.  Nothing here
touches payload bytes.
"""
import ctypes
import os
import threading
import time
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from torchrua_amd import _lib as L


class host_serial:
    """Run a torch CPU op on ONE thread.  Handing a 512 KB op to a 128-thread OpenMP team costs 10-100x its serial
    time (measured on the MI355X box: torch.sort of 65 536 lengths 20 ms vs 1.9 ms), and the sort's result does not
    depend on the thread count (SURVEY.md §8a note).  Flipping torch's thread count is process-global, so the hot
    path no longer does it: the host sort, batch_sizes and the staging copies are the library's own C / numpy code;
    this is left for the one-off self-test and the RUA_HOST_SORT=torch fallback."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        if self.n != 1:
            torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        if self.n != 1:
            torch.set_num_threads(self.n)
        return False


# ------------------------------------------------------------------ per-tensor memo
def _version(t: Tensor) -> int:
    """The tensor's version counter; inference tensors keep none (and cannot be written in place outside
    inference mode), so they memoise under a constant."""
    return 0 if t.is_inference() else t._version


def _memo_get(t: Tensor, key: str):
    memo = t.__dict__.get('_rua_memo')
    if memo is None:
        return None
    hit = memo.get(key)
    if hit is None or hit[0] != _version(t):
        return None
    return hit[1]


def _memo_put(t: Tensor, key: str, value):
    memo = t.__dict__.get('_rua_memo')
    if memo is None:
        memo = {}
        t.__dict__['_rua_memo'] = memo
    memo[key] = (_version(t), value)
    return value


def forget(t: Tensor) -> None:
    """Drop everything memoised on `t` (bench.py uses this so that no step reuses the last one's work)."""
    t.__dict__.pop('_rua_memo', None)


# ------------------------------------------------------------------ raw kernel wrappers (int64 metadata)
def exclusive_scan(x: Tensor, want_total: bool = False):
    """K1 rua_exclusive_scan_i64 — reference utils.py:16-19 (get_offsets)."""
    dev = L.require_device(x)
    lib = L.load()
    n = x.numel()
    out = torch.empty(n, dtype=torch.long, device=dev)
    total = torch.empty(2, dtype=torch.long, device=dev) if want_total else None     # [sum, #{x <= 0}] (rua.h, ABI 6)
    ws = torch.empty(lib.rua_scan_ws_elems(n), dtype=torch.long, device=dev)
    L.check(lib.rua_exclusive_scan_i64(L.ptr(x), L.ptr(out), L.ptr(total), n, L.ptr(ws), L.stream_ptr(dev)),
            'rua_exclusive_scan_i64')
    return (out, total) if want_total else out


def _as_lens(t: Tensor) -> Tensor:
    if t.dtype != torch.long or not t.is_contiguous():
        t = t.to(torch.long).contiguous()
    return t


def host_lens(token_sizes: Tensor) -> Tensor:
    """CPU copy of a length vector.  Free when the vector was built from host data (C.new / with_host_sizes attach
    the mirror); otherwise ONE blocking D2H, then memoised.
    The reference pays a sync of this kind in every size() (layout/cat.py:61-66)."""
    if not token_sizes.is_cuda:
        return token_sizes
    hit = _memo_get(token_sizes, 'host')
    if hit is not None:
        return hit
    return _memo_put(token_sizes, 'host', _read_back(token_sizes))


def _read_back(t: Tensor) -> Tensor:
    """Blocking D2H of a small vector into PINNED memory: no staging through a pageable bounce buffer, and no fresh
    pageable allocation per call (on this platform freeing one can stall the GPU queues: an munmap runs the driver's
    MMU notifier — measured 90 ms every few steps at the north-star size).  torch's caching host allocator hands the
    same pinned blocks out again once their copy is complete, which it is when this returns."""
    src = t.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    out = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    out.copy_(src, non_blocking=True)
    torch.cuda.current_stream(src.device).synchronize()
    return out


def attach_host(token_sizes: Tensor, host: Tensor) -> None:
    """Record the host copy of a device length vector.  `host` must be the library's own (nobody else writes it)."""
    _memo_put(token_sizes, 'host', host)


def private_host_copy(host: Tensor) -> Tensor:
    """A copy of the caller's host lengths that the library owns (a loader that refills its buffer through
    numpy / data_ptr does not bump `_version`, so a borrowed mirror can go stale silently).  It comes from torch's
    PINNED caching host allocator: freeing it is a free-list push, not an munmap (a fresh pageable 512 KiB block per
    step stalled the GPU queues on this platform — see _read_back), and the upload reads straight from it, so the
    copy replaces the one into a staging slot instead of adding to it."""
    if host.is_cuda:
        raise L.RuaError('with_host_sizes takes lengths that live on the HOST (got a tensor on ' + str(host.device) +
                         '); for device lengths build the container directly: C(data, token_sizes)')
    src = _as_lens(host.detach())               # (detach: numpy refuses a tensor that requires grad)
    out = torch.empty(src.shape, dtype=torch.long, pin_memory=True)
    np.copyto(out.numpy(), src.numpy())
    return out


def max_len(token_sizes: Tensor) -> int:
    h = host_lens(token_sizes)
    hit = _memo_get(token_sizes, 'max')
    if hit is not None:
        return hit
    return _memo_put(token_sizes, 'max', int(h.detach().numpy().max()) if h.numel() else 0)


def total_len(token_sizes: Tensor) -> int:
    hit = _memo_get(token_sizes, 'sum')
    if hit is not None:
        return hit
    h = host_lens(token_sizes)
    return _memo_put(token_sizes, 'sum', int(h.detach().numpy().sum()))


def dev_off(token_sizes: Tensor) -> Tensor:
    """Exclusive offsets of a device length vector (K1), memoised — and with them the scan's count of lengths <= 0
    (`dev_n_empty`): one word on the device that lets max / min / logsumexp skip tracking the reference's global `initial`
    when it reads 0 (rua_layout::bsz of a CAT layout)."""
    hit = _memo_get(token_sizes, 'off')
    if hit is not None:
        return hit
    off, total = exclusive_scan(_as_lens(token_sizes), want_total=True)
    _memo_put(token_sizes, 'n_empty_dev', total[1:2])
    return _memo_put(token_sizes, 'off', off)


def dev_n_empty(token_sizes: Tensor) -> Optional[Tensor]:
    """The one-word device tensor dev_off left behind (None when the offsets came from somewhere else)."""
    return _memo_get(token_sizes, 'n_empty_dev')


def known_no_empty(token_sizes: Optional[Tensor]) -> bool:
    """True when the HOST can tell without a device sync that every sequence holds at least one row."""
    if token_sizes is None:
        return False
    hit = _memo_get(token_sizes, 'min')
    if hit is None:
        if token_sizes.is_cuda and _memo_get(token_sizes, 'host') is None:
            return False
        h = host_lens(token_sizes)
        hit = _memo_put(token_sizes, 'min', int(h.detach().numpy().min()) if h.numel() else 1)
    return hit > 0


# ------------------------------------------------------------------ PackedSequence metadata
def pack_B(p) -> int:
    """The reference's P.size()[0] = batch_sizes.max() (layout/pack.py:12-17): the number of NON-EMPTY sequences."""
    return int(p.batch_sizes[0]) if p.batch_sizes.numel() else 0


def pack_nseq(p) -> int:
    """Sequences in the batch, zero-length ones included: what sorted/unsorted_indices and the lengths are sized
    by.  `batch_sizes[0]` only bounds the RANK of a stored row; using it as the batch size would turn every
    sequence whose index is >= the non-empty count into padding (C.pack() of lens such as [0,3,0,2])."""
    if p.unsorted_indices is not None:
        return int(p.unsorted_indices.numel())
    if p.sorted_indices is not None:
        return int(p.sorted_indices.numel())
    return pack_B(p)


def pack_boff(p) -> Tensor:
    """Device exclusive offsets of batch_sizes (reference pack.py:43-45 redoes H2D + cumsum per call)."""
    dev = p.data.device
    key = f'boff:{dev}'
    hit = _memo_get(p.batch_sizes, key)
    if hit is not None:
        return hit
    bsz = p.batch_sizes.to(device=dev, dtype=torch.long, non_blocking=True)
    _memo_put(p.batch_sizes, f'dev:{dev}', bsz)
    return _memo_put(p.batch_sizes, key, exclusive_scan(bsz))


def pack_bsz_dev(p) -> Tensor:
    pack_boff(p)
    return _memo_get(p.batch_sizes, f'dev:{p.data.device}')


def pack_lens(p) -> Tensor:
    """token_sizes of a PackedSequence in batch order (K3b) — reference core/view.py:21-25.
    Memoised on the batch_sizes OBJECT together with the unsorted_indices object it was derived with
    (identity-checked: a slice such as P.head's batch_sizes[:n] is a new object and starts clean)."""
    hit = _memo_get(p.batch_sizes, 'lens')
    if hit is not None and hit[0] is p.unsorted_indices and hit[1].device == p.data.device:
        return hit[1]
    dev = L.require_device(p.data)
    lib = L.load()
    B, T = pack_nseq(p), p.batch_sizes.numel()
    bsz = pack_bsz_dev(p)
    lens = torch.empty(B, dtype=torch.long, device=dev)
    L.check(lib.rua_lens_from_pack(L.ptr(bsz), T, L.ptr(p.unsorted_indices), B, L.ptr(lens), L.stream_ptr(dev)),
            'rua_lens_from_pack')
    # T and sum are known on the host for free
    _memo_put(lens, 'max', T)
    _memo_put(lens, 'sum', int(p.data.size(0)))
    _memo_put(p.batch_sizes, 'lens', (p.unsorted_indices, lens))
    return lens


def adopt_pack(p, lens: Tensor, boff: Tensor, bsz_dev: Tensor) -> None:
    """Record what pack() already computed so later ops on `p` do not recompute it."""
    dev = p.data.device
    _memo_put(p.batch_sizes, f'boff:{dev}', boff)
    _memo_put(p.batch_sizes, f'dev:{dev}', bsz_dev)
    # The lengths are kept as an ALIAS (same storage and version counter, another tensor object) that inherits the
    # plain-data memos: `lens` itself memoises this very batch_sizes (core._pack_meta), and batch_sizes pointing
    # back at it would close a reference cycle per pack() — cyclic garbage that forces full GC passes.
    alias = lens.detach()
    for key in ('host', 'max', 'sum', 'off'):
        hit = _memo_get(lens, key)
        if hit is not None:
            _memo_put(alias, key, hit)
    _memo_put(p.batch_sizes, 'lens', (p.unsorted_indices, alias))


def batch_sizes_from_host_lens(h: Tensor, T: int) -> Tensor:
    """batch_sizes[t] = #{b: len[b] > t} as the CPU int64 tensor PackedSequence mandates
    (reference core/view.py:55: get_mask(self).sum(dim=0).cpu()) — a histogram and a running sum on the host
    (rua_host_batch_sizes), T * 8 bytes of fresh host memory per call."""
    out = torch.empty(T, dtype=torch.long)
    if T:
        h = _as_lens(h)
        L.check(L.load().rua_host_batch_sizes(h.data_ptr(), h.numel(), T, out.data_ptr()), 'rua_host_batch_sizes')
    return out


# ------------------------------------------------------------------ the reference's host sort (core/view.py:48)
_host_sort_threads: Optional[int] = None      # None: not decided yet; 0: torch.sort; n >= 1: rua_host_sort_desc


def _selftest_inputs():
    g = torch.Generator().manual_seed(20261004)
    for n in (1, 2, 3, 16, 17, 18, 31, 33, 100, 1000, 4097, 9000, 40000):
        for hi in (1, 3, 500, 1 << 40):
            yield torch.randint(0, hi + 1, (n,), generator=g)
    n = 5000
    up = torch.arange(n)
    yield up
    yield up.flip(0)
    yield torch.zeros(n, dtype=torch.long)
    yield torch.cat([up[:n // 2], up[:n // 2].flip(0)])          # organ pipe
    yield (up * 7919) % 13                                       # few distinct values, periodic
    yield torch.cat([torch.zeros(n // 2, dtype=torch.long), torch.ones(n // 2, dtype=torch.long)])
    # a median-of-3 killer (McIlroy's adversary run against this very introsort: scripts/exp/sort_killer.cpp), 20 000
    # elements with ties: the only kind of input that exhausts the depth budget and reaches the heap-sort branch
    killer = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'selftest_sort_killer.npy')
    yield torch.from_numpy(np.load(killer).astype(np.int64))


def _host_sort_decide() -> int:
    """Pick how sorted_indices is computed on the host, once per process.

    RUA_HOST_SORT=torch keeps the reference's own call; RUA_HOST_SORT=<n> fixes the thread count.  Otherwise the
    library's reproduction of that sort (rua_host.cpp) is used IF it returns exactly torch.sort's permutation on a
    battery of inputs (tie-heavy, sorted, reversed, constant, organ-pipe, sizes around the 16-element leaf, a median-of-3 killer
    that drives it into the heap-sort branch) —
    a mismatch (another C++ runtime behind torch, say) silently keeps the reference's call.  The thread count is the
    smallest of {1, 2, 4, 8} within 5 % of the fastest on a 64 Ki-element sample (best of five runs each), so a host that
    serialises threads is not made slower."""
    env = os.environ.get('RUA_HOST_SORT', '').strip().lower()
    if env == 'torch':
        return 0
    lib = L.load()

    def ours(keys: Tensor, threads: int) -> Tensor:
        out = torch.empty_like(keys)
        L.check(lib.rua_host_sort_desc(keys.data_ptr(), keys.numel(), out.data_ptr(), threads), 'rua_host_sort_desc')
        return out

    with host_serial():
        for keys in _selftest_inputs():
            ref = torch.sort(keys, descending=True)[1]
            for threads in (1, 3):
                if not torch.equal(ours(keys, threads), ref):
                    return 0
        if env.isdigit() and int(env) >= 1:
            return min(int(env), 64)
        cpus = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        # several ranks on one host (torch.distributed.run sets LOCAL_WORLD_SIZE) each run this test and then their
        # own sorts at the same time: a rank counts only its share of the cores (parallel.bind_rank_to_cpus makes that
        # literal), so that eight ranks do not each conclude that eight threads are free
        cpus = max(1, cpus // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)))
        sample = torch.randint(8, 513, (65536,), generator=torch.Generator().manual_seed(1))
        # the best of five runs per candidate (a sum of three let one preempted run decide: a box of round 4 settled on
        # 2 threads, 0.24 ms, where 4 take 0.19), then the SMALLEST thread count within 5 % of the fastest
        times = {}
        for threads in (1, 2, 4, 8):
            if threads > max(1, cpus):
                break
            ours(sample, threads)
            best_run = None
            for _ in range(5):
                t0 = time.perf_counter()
                ours(sample, threads)
                dt = time.perf_counter() - t0
                best_run = dt if best_run is None or dt < best_run else best_run
            times[threads] = best_run
        fastest = min(times.values())
        return min(t for t, dt in times.items() if dt <= 1.05 * fastest)


def host_sort_threads() -> int:
    """Threads the library's reproduction of the host sort runs on (0: the reference's own torch.sort is kept)."""
    global _host_sort_threads
    if _host_sort_threads is None:
        _host_sort_threads = _host_sort_decide()
    return _host_sort_threads


def host_sort_is_native() -> bool:
    return host_sort_threads() >= 1


def host_sort_desc(host: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """sorted_indices = torch.sort(host, descending=True)[1] (core/view.py:48) — the same permutation, tie order
    included — into `out` if given (e.g. a pinned staging slot)."""
    global _host_sort_threads
    if _host_sort_threads is None:
        _host_sort_threads = _host_sort_decide()
    host = _as_lens(host.detach())
    if _host_sort_threads == 0:
        with host_serial():
            index = torch.sort(host, descending=True)[1]
        if out is None:
            return index
        out.copy_(index)
        return out
    if out is None:
        out = torch.empty(host.shape, dtype=torch.long)
    L.check(L.load().rua_host_sort_desc(host.data_ptr(), host.numel(), out.data_ptr(), _host_sort_threads),
            'rua_host_sort_desc')
    return out


# ------------------------------------------------------------------ rua_layout descriptors
def known_max_len(token_sizes: Optional[Tensor]) -> Optional[int]:
    """max(token_sizes) if the host can tell WITHOUT a device sync (memoised, or a host mirror exists)."""
    if token_sizes is None:
        return None
    hit = _memo_get(token_sizes, 'max')
    if hit is not None:
        return hit
    if not token_sizes.is_cuda or _memo_get(token_sizes, 'host') is not None:
        return max_len(token_sizes)
    return None


class Lay:
    """A rua_layout plus the tensors its pointers borrow (kept alive with it)."""
    __slots__ = ('c', 'keep', 'kind', 'n_rows', 'B', 'max_len', 'heavy_tail', '_no_empty')

    def __init__(self, keep: List[Optional[Tensor]], max_len: Optional[int] = None, **fields):
        self.c = L.RuaLayout(**fields)
        self.keep = keep
        self.kind = fields['kind']
        self.n_rows = fields['n_rows']
        self.B = fields['B']
        self.max_len = max_len      # longest sequence, when the host knows it for free
        self.heavy_tail = False     # the buckets of a scatter_*: sizes counted on the device, ONE hot bucket is ordinary
        self._no_empty = False      # bool, or a callable that decides on first use (only max / min / logsumexp ask)

    def ref(self):
        return ctypes.byref(self.c)

    @property
    def no_empty(self) -> bool:
        """The HOST can prove, without a device sync, that every sequence holds at least one row (RUA_OP_NO_EMPTY)."""
        if callable(self._no_empty):
            self._no_empty = bool(self._no_empty())
        return self._no_empty


def lay_cat(lens: Optional[Tensor], B: int, n_rows: int, len_add: int = 0) -> Lay:
    """CAT rows; `lens=None` means every sequence has exactly `len_add` rows."""
    if lens is None:
        return Lay([], max_len=len_add, kind=L.CAT, n_rows=n_rows, B=B, len_add=len_add)
    lens = _as_lens(lens)
    off = dev_off(lens)
    mx = known_max_len(lens)
    longest = None if mx is None else mx + len_add
    # (bsz of a CAT layout: the device's own count of empty sequences, when the offsets came from dev_off's scan)
    ne = dev_n_empty(lens) if len_add == 0 else None
    lay = Lay([lens, off, ne], max_len=longest, kind=L.CAT, n_rows=n_rows, B=B, lens=L.ptr(lens), len_add=len_add,
              off=L.ptr(off), T_log=longest or 0, bsz=L.ptr(ne))   # (T_log of a CAT layout: the longest sequence, 0 = unknown)
    lay._no_empty = (lambda: known_no_empty(lens)) if len_add >= 0 else False
    return lay


def lay_padded(kind: int, lens: Optional[Tensor], B: int, T_phys: int, T_log: int, len_add: int = 0,
               with_off: bool = False) -> Lay:
    keep: List[Optional[Tensor]] = []
    f = dict(kind=kind, n_rows=B * T_phys, B=B, T_phys=T_phys, T_log=T_log, len_add=len_add)
    if lens is not None:
        lens = _as_lens(lens)
        keep.append(lens)
        f['lens'] = L.ptr(lens)
        if with_off:  # needed only to ENUMERATE tokens (ptr()/idx())
            off = dev_off(lens)
            keep.append(off)
            f['off'] = L.ptr(off)
    mx = len_add if lens is None else known_max_len(lens)
    lay = Lay(keep, max_len=mx if lens is None or mx is None else mx + len_add, **f)
    lay._no_empty = (len_add > 0) if lens is None else ((lambda: known_no_empty(lens)) if len_add >= 0 else False)
    return lay


NARROW_ROW_BYTES = 64      # rows up to this get the tile table (rua_move.hip: TILE_MAX_ROW_BYTES)
TILE_MIN_LIVE_INV = 4       # ... when at least one cell in this many is a live token


# out of a PackedSequence (P.cat(), P.left()) the batch-major side is the one WRITTEN: taller tiles make its runs longer
_FROM_PACK_SHAPES = {8: (7, 4), 4: (7, 5)} if os.environ.get('RUA_TILE_FROM_PACK', '1') != '0' else {}


_TALL_BOTH_WAYS = False        # (developer A/B: scripts/exp/from_pack_ab.py)


def tile_shape_log2(row_bytes: int, from_pack: bool = False) -> Tuple[int, int]:
    """(log2 time steps, log2 ranks) of a (rank x time) tile by row width (rua_move.hip: pack_tile_lds_kernel).  [r5] Rows
    of ONE vector below 16 bytes — 1-D payloads of 8 / 4 / 2 / 1-byte elements — get more ranks (and steps) per tile, so
    that a tile still carries 16 KiB and both sides still move runs of 128 .. 512 bytes: 32 x 64, 64 x 64, 64 x 128,
    128 x 128.  (The kernel falls back to the row mover when the payload's address is less aligned than its rows.)"""
    narrow = {8: (6, 5), 4: (6, 6), 2: (7, 6), 1: (7, 7)}.get(row_bytes)
    if (from_pack or _TALL_BOTH_WAYS) and row_bytes in _FROM_PACK_SHAPES:
        narrow = _FROM_PACK_SHAPES[row_bytes]
    if narrow is not None:
        return narrow
    return (6 if row_bytes <= 16 else 5 if row_bytes <= 32 else 4), 4


TILE_FULL_GRID = 1 << 24      # rua_layout::tile_t_log2: the tiles cover the whole (sequence x step) grid of a padded destination
TILE_STEP_ROWS = 1 << 25      # ... the table counts tiles of ONE time step x (1 << trl) ranks (a roll inside a PackedSequence)
STEP_TILE_BYTES = 16 << 10
STEP_MAX_ROW_BYTES = 32       # wider rows roll at 5.8 - 6.3 TB/s on the row mover already


def lay_pack_steps(p, row_bytes: int, shift: int) -> Optional['Lay']:
    """The PACK layout of `p` with the table of one-step tiles that `P.roll(shift)` moves a PackedSequence of narrow rows
    on (rua_move.hip: pack_roll_steps_kernel) — or None when that is not the way to go: rows wider than 32 bytes or not
    a shift that makes most ranks wrap, or steps so short that the tiles would be mostly empty."""
    T = p.batch_sizes.numel()
    if not 0 < row_bytes <= STEP_MAX_ROW_BYTES or T == 0 or abs(shift) * 8 > T:
        return None
    trl = max(4, (STEP_TILE_BYTES // row_bytes).bit_length() - 1)
    dev = p.data.device
    key = f'tiling:{dev}:0:{trl}:steps'
    t = _memo_get(p.batch_sizes, key)
    if t is None:
        t = _memo_put(p.batch_sizes, key, PackTiling(p.batch_sizes, pack_bsz_dev(p), dev, 0, trl))
    n_rows = int(p.data.size(0))
    if n_rows * TILE_MIN_LIVE_INV < (t.n_tiles << trl):
        return None
    lay = lay_pack(p)
    fields = {name: getattr(lay.c, name) for name, _ in L.RuaLayout._fields_}
    fields.update(bsz=L.ptr(t.bsz), tile_start=L.ptr(t.tile_start), n_tchunks=t.n_tchunks, n_tiles=t.n_tiles,
                  tile_t_log2=(trl << 8) | TILE_STEP_ROWS)
    out = Lay(list(lay.keep) + [t.bsz, t.tile_start], max_len=lay.max_len, **fields)
    out._no_empty = lay._no_empty
    return out


def tile_line_rows(row_bytes: int) -> int:
    """R = rows per 128-byte line when the rows divide one (16 / 32 / 64-byte rows: 8 / 4 / 2), else 1.  With R > 1 the
    narrow-row tile kernel gives every rank its own time origin so that its runs on the batch-major side are whole
    lines (rua_move.hip: TileTables); the tile table below is then built for windows shifted by up to R - 1 steps."""
    return 128 // row_bytes if row_bytes in (16, 32, 64) else 1


class PackTiling:
    """Tile table for the narrow-row C/L/R <-> P kernel, derived on the host from batch_sizes (a CPU tensor by
    PackedSequence's contract): tile_start[c] = number of rank tiles before time chunk c (chunks of 1 << ttl steps,
    tiles of 1 << trl ranks).  line_rows = R > 1: a rank's window may begin up to R - 1 steps before the chunk, so
    chunk c takes the ranks alive at step c * TT - (R - 1), and there are ceil((T + R - 1) / TT) chunks."""
    __slots__ = ('bsz', 'tile_start', 'n_tchunks', 'n_tiles', 'code')

    def __init__(self, batch_sizes: Tensor, bsz_dev: Tensor, dev: torch.device, ttl: int, trl: int, line_rows: int = 1):
        # numpy on the (CPU, by PackedSequence's contract) batch_sizes: a handful of torch CPU ops on a few dozen
        # elements cost ~30 us EACH on the GPU box's 128-thread host build
        bs = batch_sizes.numpy()
        tt, extra = 1 << ttl, max(line_rows, 1) - 1
        n_chunks = (bs.size + extra + tt - 1) // tt if bs.size else 0
        first = np.maximum(np.arange(n_chunks, dtype=np.int64) * tt - extra, 0)
        counts = (bs[first] + ((1 << trl) - 1)) >> trl
        start = np.zeros(counts.size + 1, dtype=np.int64)
        np.cumsum(counts, out=start[1:])
        self.n_tchunks = int(counts.size)
        self.n_tiles = int(start[-1])
        self.tile_start = to_device_async(torch.from_numpy(start), dev)
        self.bsz = bsz_dev
        self.code = ttl | (trl << 8) | ((line_rows if line_rows > 1 else 0) << 16)


def pack_tiling(p, ttl: int, trl: int, line_rows: int = 1) -> 'PackTiling':
    dev = p.data.device
    key = f'tiling:{dev}:{ttl}:{trl}:{line_rows}'
    hit = _memo_get(p.batch_sizes, key)
    if hit is not None:
        return hit
    return _memo_put(p.batch_sizes, key, PackTiling(p.batch_sizes, pack_bsz_dev(p), dev, ttl, trl, line_rows))


def lay_pack(p, lens: Optional[Tensor] = None, len_add: int = 0, boff: Optional[Tensor] = None,
             T: Optional[int] = None, n_rows: Optional[int] = None, row_bytes: Optional[int] = None,
             full_grid_T: Optional[int] = None, from_pack: bool = False) -> Lay:
    lens = pack_lens(p) if lens is None else _as_lens(lens)
    boff = pack_boff(p) if boff is None else boff
    T = p.batch_sizes.numel() if T is None else T
    n_rows = int(p.data.size(0)) if n_rows is None else n_rows
    keep = [lens, boff, p.sorted_indices, p.unsorted_indices]
    extra = {}
    if T == p.batch_sizes.numel():      # the device copy of batch_sizes (made together with boff)
        bsz = pack_bsz_dev(p)
        keep.append(bsz)
        extra['bsz'] = L.ptr(bsz)
    if (full_grid_T is not None and row_bytes is not None and 0 < row_bytes <= NARROW_ROW_BYTES
            and T == p.batch_sizes.numel() and len_add == 0 and T > 0):
        # [r5] the SOURCE of a pad at narrow rows (P.left() / P.right()): tiles over the destination's whole
        # (sequence x step) grid — tokens and fill in one pass; no table: every chunk holds ceil(B / ranks per tile) tiles
        ttl, trl = tile_shape_log2(row_bytes)          # (the taller tiles lose 1-6 % here: r5j/from_pack_ab.txt)
        nseq = pack_nseq(p)
        n_tchunks = (max(full_grid_T, T) + (1 << ttl) - 1) >> ttl
        extra.update(n_tchunks=n_tchunks, n_tiles=n_tchunks * ((nseq + (1 << trl) - 1) >> trl),
                     tile_t_log2=ttl | (trl << 8) | TILE_FULL_GRID)
    elif row_bytes is not None and 0 < row_bytes <= NARROW_ROW_BYTES and T == p.batch_sizes.numel() and len_add == 0:
        ttl, trl = tile_shape_log2(row_bytes, from_pack)
        t = pack_tiling(p, ttl, trl, tile_line_rows(row_bytes))       # narrow rows: hand the (rank x time) tile table to the mover ...
        # ... unless the tiles would be mostly dead cells: one giant sequence among short ones (its tail is one live rank
        # in sixteen), or a batch of singletons (one live step in 16..64).  Below a quarter of live cells the generic
        # mover — a row per lane, nothing dead — is 2-7 x faster (profiles/r04_shape_cliffs.txt)
        if n_rows * TILE_MIN_LIVE_INV >= (t.n_tiles << (ttl + trl)):
            keep += [t.bsz, t.tile_start]
            extra.update(bsz=L.ptr(t.bsz), tile_start=L.ptr(t.tile_start), n_tchunks=t.n_tchunks, n_tiles=t.n_tiles,
                         tile_t_log2=t.code)
    lay = Lay(keep, max_len=T, kind=L.PACK, n_rows=n_rows, B=pack_nseq(p),
              lens=L.ptr(lens), len_add=len_add, boff=L.ptr(boff), T=T, sorted=L.ptr(p.sorted_indices),
              unsorted=L.ptr(p.unsorted_indices), **extra)
    # batch_sizes[0] counts the sequences that hold a row (a host tensor): all of them, unless some are empty
    lay._no_empty = len_add == 0 and T == p.batch_sizes.numel() and T > 0 and pack_B(p) == pack_nseq(p)
    return lay


class _StagingRing:
    """A small ring of pinned host buffers for the per-call metadata uploads (lengths, sorted_indices).

    `tensor.pin_memory()` per call is wrong here: while the GPU is busy the caching host allocator
    cannot recycle blocks whose copy events are still pending, so every call pays a fresh
    hipHostMalloc (tens of ms on the MI355X box).  The ring reuses SLOTS buffers; before a slot is
    reused the host waits for that slot's last copy, which also bounds how far the host may run
    ahead of the stream (SLOTS uploads = about 16 pack() calls: deep enough to ride out host hiccups)."""
    SLOTS = 32
    SIDE_MIN_ELEMS = 1024    # smaller uploads stay on the current stream (see upload)

    def __init__(self):
        self.bufs = [None] * self.SLOTS
        self.events = [None] * self.SLOTS
        self.busy = [False] * self.SLOTS     # reserved, not yet committed (a host thread is filling it)
        self.i = 0
        self.side = None      # the upload stream of this device
        self.lock = threading.Lock()         # two host threads must never be handed the same slot

    def reserve(self, shape, dtype: torch.dtype) -> Tuple[int, Tensor]:
        """Claim the next slot and return (slot, pinned tensor of `shape`) for the caller to fill."""
        with self.lock:
            i = self.i
            for _ in range(self.SLOTS):
                if not self.busy[i]:
                    break
                i = (i + 1) % self.SLOTS
            else:
                raise RuntimeError('torchrua_amd: every staging slot is being filled by another thread')
            self.busy[i] = True
            self.i = (i + 1) % self.SLOTS
        if self.events[i] is not None:
            self.events[i].synchronize()
            self.events[i] = None
        n = 1
        for d in shape:
            n *= d
        nbytes = n * dtype.itemsize
        buf = self.bufs[i]
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)
            self.bufs[i] = buf
        return i, buf[:nbytes].view(dtype).view(shape)

    def commit(self, i: int, staged: Tensor, dev: torch.device) -> Tensor:
        """Enqueue the H2D of a slot filled through reserve()."""
        try:
            return self._commit(i, staged, dev)
        finally:
            self.busy[i] = False

    def _commit(self, i: int, staged: Tensor, dev: torch.device) -> Tensor:
        out, ev = self.h2d(staged, dev)
        self.events[i] = ev
        return out

    def h2d(self, staged: Tensor, dev: torch.device):
        """Enqueue the H2D of a PINNED host tensor; returns (device tensor, event that follows the copy)."""
        cur = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        if staged.numel() < self.SIDE_MIN_ELEMS or torch.cuda.is_current_stream_capturing():
            out = torch.empty(staged.shape, dtype=staged.dtype, device=dev)
            out.copy_(staged, non_blocking=True)
            ev.record(cur)
        else:
            # (Vectors of a few dozen lengths stay on the current stream: there the cross-stream hand-off costs more
            # host time — C.new(xs).left() at cfg1 59 vs 77 us — than the overlap can win; from mid sizes on it is the
            # other way round — cfg2 steady state 0.41 vs 0.48 ms/step.)
            # The copy runs on its OWN stream: it depends on nothing the compute stream holds (the source is host
            # memory written just above), so the copy engine works while the previous step's kernels still run and
            # the consumer finds its input ready, instead of a copy -> kernel hand-off sitting in the middle of the
            # compute queue.  The destination comes from the side stream's pool (a block freed on the compute stream
            # may still be read by a pending kernel) and is handed to the compute stream with record_stream.
            side = self.side
            if side is None:
                side = self.side = torch.cuda.Stream(dev)
            # (set_stream twice instead of the `torch.cuda.stream` context manager: a third of its cost, and this
            # runs twice per pack())
            torch.cuda.set_stream(side)
            try:
                out = torch.empty(staged.shape, dtype=staged.dtype, device=dev)
                out.copy_(staged, non_blocking=True)
                ev.record(side)
            finally:
                torch.cuda.set_stream(cur)
            cur.wait_event(ev)
            out.record_stream(cur)
        return out, ev

    def upload(self, host: Tensor, dev: torch.device) -> Tensor:
        i, staged = self.reserve(host.shape, host.dtype)
        # a plain memcpy on the calling thread: torch's copy_ hands a 512 KiB vector to its whole OpenMP team
        # (128 threads on the GPU box: 10-100x the serial time), and flipping torch.set_num_threads around it — what
        # round 1 did — is a process-global side effect
        try:
            np.copyto(staged.numpy(), host.detach().numpy())
        except BaseException:
            self.busy[i] = False
            raise
        return self.commit(i, staged, dev)


_rings = {}


def _ring(dev: torch.device) -> _StagingRing:
    ring = _rings.get(dev)
    if ring is None:
        ring = _rings[dev] = _StagingRing()
    return ring


def to_device_async(host: Tensor, dev: torch.device) -> Tensor:
    """Enqueue the H2D of a small contiguous host vector (on the device's upload stream; the current stream waits
    for it) without blocking the host."""
    if dev.type != 'cuda' or host.numel() == 0 or not host.is_contiguous():
        return host.to(dev)
    return _ring(dev).upload(host, dev)


def pinned_to_device_async(pinned: Tensor, dev: torch.device) -> Tensor:
    """H2D of a pinned host tensor the caller keeps alive (torch's host allocator does not hand a pinned block out
    again while a copy from it is pending), on the upload stream."""
    if pinned.numel() == 0:
        return pinned.to(dev)
    return _ring(dev).h2d(pinned, dev)[0]


def sorted_indices_to_device(host_lens_: Tensor, dev: torch.device, stream_idle: bool = False) -> Tensor:
    """The reference's host sort (core/view.py:48), written straight into a pinned staging slot and uploaded."""
    n = host_lens_.numel()
    if dev.type != 'cuda' or n == 0:
        return host_sort_desc(host_lens_).to(dev)
    if stream_idle:
        # device-only lengths: the compute stream was synchronised a moment ago and the GPU idles until the mover is
        # launched, so every host microsecond here shows.  No ring slot, no events, no side stream: a block from torch's
        # pinned host cache, the sort straight into it, ONE async copy on the current stream (the host allocator keeps
        # the block until that copy is done).
        staged = torch.empty(n, dtype=torch.long, pin_memory=True)
        host_sort_desc(host_lens_, out=staged)
        return staged.to(dev, non_blocking=True)
    ring = _ring(dev)
    i, staged = ring.reserve((n,), torch.long)
    try:
        host_sort_desc(host_lens_, out=staged)
    except BaseException:
        ring.busy[i] = False
        raise
    return ring.commit(i, staged, dev)


def lay_list(bptr: Optional[Tensor], tptr: Tensor) -> Lay:
    tptr = _as_lens(tptr)
    bptr = None if bptr is None else _as_lens(bptr)
    return Lay([bptr, tptr], kind=L.LIST, n_rows=tptr.numel(), B=0, bptr=L.ptr(bptr), tptr=L.ptr(tptr))


def lay_flat(n_rows: int) -> Lay:
    """A storage of n_rows rows seen as ONE left-aligned sequence (target of flat row gathers)."""
    return Lay([], kind=L.LEFT, n_rows=n_rows, B=1, T_phys=n_rows, T_log=n_rows, len_add=n_rows)


def hidden_of(data: Tensor, lead: int) -> Tuple[int, ...]:
    return tuple(data.shape[lead:])


def row_bytes(data: Tensor, lead: int) -> int:
    n = 1
    for d in data.shape[lead:]:
        n *= d
    return n * data.element_size()


# ------------------------------------------------------------------ long-sequence splitting of the reducer
SPLIT_ROWS = 4096            # upper bound on rows per part
SPLIT_MIN_BYTES = 64 << 10   # lower bound on a part's bytes per wave: its 4-KiB fp32 partial stays a few % of traffic
SPLIT_MIN_ROWS = 32
FILL_WAVES = 8192            # 256 CUs x 32 waves
ENOUGH_UNITS = 4096          # (sequence, column chunk) units that keep every SIMD busy without splitting
WAVE_RATE = 4e9              # bytes/s ONE wave streams (8 KiB in flight / ~2 us; profiles/r01_skew.txt)
STREAM_RATE = 5e12           # bytes/s the whole chip reads through the reducer
SPLIT_FIXED_S = 30e-6        # what arming costs: one memset, a tail and a combine launch
ARM_ALWAYS_BYTES_PLAIN = 1 << 30
ARM_ALWAYS_BYTES = 64 << 20  # scatter_* payloads from here on always arm the split (bucket sizes live on the device)


def reduce_split_rows(lay: Lay, row_bytes: int = 1024, team_ok: bool = True, tail_ok: bool = True) -> int:
    """Rows per part for rua_segment_reduce, or 0 (= one wave streams each whole sequence).

    Splitting pays only when the longest sequence would show: one wave walks a sequence at ~WAVE_RATE, the balanced
    chip needs n_rows*row_bytes/STREAM_RATE for everything, so a sequence is a problem when
    len * min(row_bytes, 1 KiB) / WAVE_RATE exceeds ~3/4 of that (`ideal_rows`) plus the fixed cost of the
    machinery (`fixed_rows`).  Measured on the mid-size BASELINE shapes (profiles/r01_mid_sizes.txt): splitting sequences that
    do not need it halves the rate.  The part size fills the chip (n_rows / 8192) when there are few units, and is
    raised to that threshold when there are plenty, so that only real outliers are cut.  When the host does not know
    the longest sequence (device-only lengths) the machinery is armed where a tail could matter: long average
    sequences, few of them, or the buckets of a scatter_* (`lay.heavy_tail`) from 64 MB of payload on.  team_ok=False: the caller's kernel has no wave teams (the backward walk, the fused
    pack + reduce), so a unit streams at the single-wave rate.  tail_ok=False: rows of 8 (mod 16) bytes will NOT take
    the 16-byte-lane path there (include_self == 1, or a payload that is not 8-byte aligned: the launcher's `tail_ok`),
    so no team either."""
    n = lay.n_rows
    rb_unit = max(1, min(int(row_bytes), 1024))        # wider rows: 4 KiB per wave, 4x the loads in flight
    n_chunks = -(-int(row_bytes) // (1024 if row_bytes <= 1024 else 4096)) if row_bytes > 0 else 1
    # few-but-long units are shared by a TEAM of 2 or 4 waves (seg_reduce_team_kernel; the same rule as the C side:
    # rows up to 1 KiB on the vector path, at most 16 384 units, >= 4 row groups per wave): a unit then streams
    # 2-4x as fast, and splitting — three launches and a pass over fp32 partials — is for real outliers only
    team = 1
    if team_ok and 0 < row_bytes <= 1024 and (row_bytes % 16 == 0 or (row_bytes % 8 == 0 and tail_ok)):
        team = L.load().rua_reduce_team_waves(n, max(lay.B, 1), int(row_bytes))      # the launcher's own rule
    wave_rate = WAVE_RATE * team
    ideal_rows = int(0.75 * n * row_bytes / STREAM_RATE * wave_rate / rb_unit)   # rows a unit walks in 3/4 of the balanced time
    fixed_rows = int(SPLIT_FIXED_S * wave_rate / rb_unit)
    part_min = max(SPLIT_MIN_ROWS, SPLIT_MIN_BYTES // rb_unit)
    part = max(part_min, min(SPLIT_ROWS, n // FILL_WAVES))
    if max(lay.B, 1) * n_chunks >= ENOUGH_UNITS:
        part = max(part, min(SPLIT_ROWS, ideal_rows + fixed_rows))
    if n <= part:
        return 0
    if lay.max_len is not None:
        # worth it when the longest walk exceeds what remains after splitting (a part, or the balanced time) + the fixed cost
        return part if lay.max_len > max(part, ideal_rows) + fixed_rows else 0
    # (the buckets of a scatter_* — `heavy_tail` — are counted on the device and a skewed histogram with ONE hot bucket
    # is the ordinary case there; an unsplit hot bucket is a cliff, not a slope — a third of 17 M rows in one of 100 000
    # buckets: 171 ms / 1.15 s at 128-byte / 1-KiB rows against 0.9 / 3.5 ms split — while arming costs 5-7 us when
    # nothing is long (profiles/r04_skew.txt): from 64 MB of payload on they always arm the machinery)
    armed = n >= 256 * max(lay.B, 1) or lay.B < 1024
    # (any other layout with device-only lengths: from 1 GB on, where the 5-7 us are below 3 % of the call)
    if n * max(1, int(row_bytes)) >= (ARM_ALWAYS_BYTES if getattr(lay, 'heavy_tail', False) else ARM_ALWAYS_BYTES_PLAIN):
        armed = True
    return part if armed else 0
