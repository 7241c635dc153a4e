"""Where a LARGE output of the row mover is put — OPT-IN, off by default.

On the MI355X the time of a multi-GB move depends on WHICH allocations it runs between: the buffers of a process fall
into a few classes (profiles/r02_placement.txt), and a move whose source and destination sit in the same class is ~6 %
slower than one between classes — the same kernel, the same bytes (north-star pack: 6.05 vs 5.65 ms).  torch's caching
allocator hands out whichever cached block fits, so a pipeline that alternates between two output blocks runs at the mix.

When switched on (RUA_PLACEMENT=1, or `enable()`), the mover learns from its own launches which CACHED block is good
for which source and prefers it.  Round 2 shipped this on by default and let it allocate while it explored; on a fresh
box that stalled a caller for 0.7 s on a new shape.  The rules now:

  * it never asks the device for anything: no hipMalloc on its behalf, no `mem_get_info`, no `memory_stats` (all three
    wait for the device).  It only asks torch's caching allocator for another block while it holds an offer, and only
    when it KNOWS another block of exactly this size is sitting in the cache — a block it handed out earlier whose
    storage has since died (weak references to the storages it handed out; no allocator query);
  * should the allocator nevertheless answer with a block never met before (the cache was flushed behind its back, the
    block went to another stream's pool), choosing is switched off for that size for good: growth by one block, once;
  * while choosing it holds at most MAX_EXTRA_BLOCKS offers beyond the one it returns (default 1) and every extra
    `torch.empty` is wrapped: an OutOfMemoryError returns the first offer and freezes the size;
  * a caller who WANTS more blocks to choose from says so: `warm(nbytes, device, blocks)` puts that many blocks of the
    size into the cache, up front, at the caller's expense.  bench.py does not.

Every launch into a large output (>= MIN_BYTES) is bracketed by two events on its stream; finished pairs are read back
later, without ever waiting, and the time is filed under (what moved, source block, output block).  Nothing here
changes a result: only which cached block the output lives in.
"""
import os
import threading
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor
from torch.multiprocessing.reductions import StorageWeakRef

ENABLED = os.environ.get('RUA_PLACEMENT', '0') == '1'
MIN_BYTES = int(os.environ.get('RUA_PLACEMENT_MIN_BYTES', 2 << 30))
TOLERANCE = 0.015          # a block within this of the best cost for its source is taken as it comes (launches between
                           # one pair of blocks repeat to ~0.3 %)
MAX_EXTRA_BLOCKS = int(os.environ.get('RUA_PLACEMENT_MAX_EXTRA', 1))   # offers held beyond the one returned
MIN_SAMPLES = 2            # a block is only called slow after this many launches (the first one into fresh memory
                           # also pays for its page tables)
PENDING_MAX = 256
TABLE_MAX = 4096           # (source, output) pairs remembered; beyond that everything is learned afresh
_lock = threading.Lock()

Key = Tuple[str, int, int, int]                      # (move name, output bytes, source bytes, source storage)
Size = Tuple[int, int]                               # (device index, bytes)
_times: Dict[Tuple[Key, int], Tuple[float, int]] = {}   # (key, output storage) -> (fastest ms, launches timed)
_pending: List[Tuple[Key, int, torch.cuda.Event, torch.cuda.Event]] = []
_tried: Dict[Key, set] = {}                          # output blocks launched into, per key (timed or still pending)
_blocks: Dict[Size, Dict[int, Optional[StorageWeakRef]]] = {}   # blocks met, with the storage last handed out on each
_frozen: Dict[Size, bool] = {}                       # choosing switched off for this size
_readers: Dict[Tuple[str, int], Dict[int, float]] = {}   # (reader kernel, bytes) -> {block: fastest ms}
stats = {'timed': 0, 'rejected': 0, 'taken_good': 0, 'taken_untimed': 0, 'settled': 0, 'frozen': 0, 'oom': 0}


def enable(on: bool = True) -> None:
    global ENABLED
    ENABLED = bool(on)


def _base(t: Tensor) -> int:
    return t.untyped_storage().data_ptr()


def _harvest() -> None:
    """File the launches that have finished; never waits."""
    if len(_times) > TABLE_MAX:
        _times.clear()
        _tried.clear()
        _readers.clear()
    keep = []
    for item in list(_pending):
        key, ob, e0, e1 = item
        if not e1.query():
            keep.append(item)
            continue
        ms = e0.elapsed_time(e1)
        stats['timed'] += 1
        if key[0].startswith('read:'):
            table = _readers.setdefault((key[0], key[1]), {})
            table[ob] = min(ms, table.get(ob, float('inf')))
            continue
        old = _times.get((key, ob))
        _times[(key, ob)] = (ms, 1) if old is None else (min(ms, old[0]), old[1] + 1)
    _pending[:] = keep[-PENDING_MAX:]


def _read_penalty(ob: int, nbytes: int) -> float:
    """ms a block costs its readers over the best block of its size, as far as readers have been timed on both."""
    worst = 0.0
    for (_name, n), table in _readers.items():
        if n == nbytes and ob in table and len(table) > 1:
            worst = max(worst, table[ob] - min(table.values()))
    return worst


def _cost(key: Key, ob: int) -> Optional[float]:
    hit = _times.get((key, ob))
    if hit is None or hit[1] < MIN_SAMPLES:
        return None
    return hit[0] + _read_penalty(ob, key[1])


def _verdict(key: Key, ob: int) -> Optional[float]:
    """The block's cost for this source when it is KNOWN to be worse than the best block tried by more than
    TOLERANCE, else None (good, or not timed enough)."""
    cost = _cost(key, ob)
    if cost is None:
        return None
    best = min((c for c in (_cost(key, b) for b in _tried.get(key, ())) if c is not None), default=cost)
    return cost if cost > best * (1.0 + TOLERANCE) else None


def _hand_out(size: Size, t: Tensor) -> Tensor:
    _blocks.setdefault(size, {})[_base(t)] = StorageWeakRef(t.untyped_storage())
    return t


def _cached_better(key: Key, size: Size, held: List[Tensor], than: float) -> bool:
    """Is a block of this size known to sit in the allocator's cache (handed out earlier, its storage dead since) that
    is not known to cost `than` or more for this source?  Pure bookkeeping: the allocator is not asked."""
    mine = {_base(t) for t in held}
    for addr, ref in _blocks.get(size, {}).items():
        if addr in mine or ref is None or not ref.expired():
            continue
        cost = _cost(key, addr)
        if cost is None or cost < than:
            return True
    return False


def key_for(name: str, out_bytes: int, src: Tensor) -> Optional[Key]:
    if not ENABLED or out_bytes < MIN_BYTES or torch.cuda.is_current_stream_capturing():
        return None
    return (name, out_bytes, src.numel() * src.element_size(), _base(src))


def empty_for(shape, dtype: torch.dtype, dev: torch.device, key: Key) -> Tensor:
    """torch.empty(shape) for an output the mover is about to fill — in a cached block that is not known to be slow
    for this source, if the cache is known to hold one."""
    with _lock:
        return _empty_for(shape, dtype, dev, key)


def _empty_for(shape, dtype: torch.dtype, dev: torch.device, key: Key) -> Tensor:
    _harvest()
    size: Size = (dev.index, key[1])
    out = torch.empty(shape, dtype=dtype, device=dev)
    slow = _verdict(key, _base(out))
    if slow is None or _frozen.get(size, False):
        stats['taken_good' if (key, _base(out)) in _times else 'taken_untimed'] += 1
        return _hand_out(size, out)
    held: List[Tuple[float, Tensor]] = [(slow, out)]
    while len(held) <= MAX_EXTRA_BLOCKS and _cached_better(key, size, [t for _, t in held], min(c for c, _ in held)):
        try:
            nxt = torch.empty(shape, dtype=dtype, device=dev)      # the held offers are alive: this is another block
        except torch.OutOfMemoryError:
            _frozen[size] = True
            stats['oom'] += 1
            break
        if _base(nxt) not in _blocks.get(size, {}):
            # the allocator grew (or re-carved its cache) to serve this: stop choosing for this size
            _frozen[size] = True
            stats['frozen'] += 1
            _hand_out(size, held[0][1])       # (the first offer goes back to the cache; remember it is there)
            return _hand_out(size, nxt)
        cost = _verdict(key, _base(nxt))
        if cost is None:
            stats['rejected'] += len(held)
            stats['taken_good' if (key, _base(nxt)) in _times else 'taken_untimed'] += 1
            for _, t in held:
                _hand_out(size, t)            # dies on return: back in the cache
            return _hand_out(size, nxt)
        held.append((cost, nxt))
    held.sort(key=lambda x: x[0])
    stats['rejected'] += len(held) - 1
    stats['settled'] += 1
    for _, t in held[1:]:
        _hand_out(size, t)
    return _hand_out(size, held[0][1])


def warm(nbytes: int, device=None, blocks: int = 3) -> int:
    """Opt-in, at the caller's expense: put `blocks` blocks of `nbytes` into torch's allocator cache so that
    empty_for has something to choose from.  Returns how many it got (an OutOfMemoryError ends it early)."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:                      # 'cuda' without an index: tensors carry one, and so must the bookkeeping
        dev = torch.device('cuda', torch.cuda.current_device())
    got: List[Tensor] = []
    with _lock:
        for _ in range(max(0, blocks)):
            try:
                got.append(torch.empty(nbytes, dtype=torch.uint8, device=dev))
            except torch.OutOfMemoryError:
                break
        for t in got:
            _hand_out((dev.index, nbytes), t)
    return len(got)


def begin(key: Key, out: Tensor, stream: torch.cuda.Stream):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    with _lock:
        _tried.setdefault(key, set()).add(_base(out))
    return key, _base(out), e0


def end(token, stream: torch.cuda.Stream) -> None:
    key, ob, e0 = token
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(stream)
    with _lock:
        _pending.append((key, ob, e0, e1))


def reader_begin(name: str, data: Tensor, stream: torch.cuda.Stream):
    """Bracket a kernel that READS a block this module placed (the reduce over a PackedSequence that pack() just made):
    its time goes into the block's cost the next time the mover chooses."""
    if not ENABLED:
        return None
    nbytes = data.numel() * data.element_size()
    if nbytes < MIN_BYTES or torch.cuda.is_current_stream_capturing():
        return None
    known = _blocks.get((data.device.index, nbytes))
    ob = _base(data)
    if not known or ob not in known:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    return ('read:' + name, nbytes, 0, 0), ob, e0


def forget() -> None:
    """Drop everything learned (call after torch.cuda.empty_cache(): a storage address may come back on other memory)."""
    with _lock:
        _times.clear()
        _pending.clear()
        _tried.clear()
        _blocks.clear()
        _frozen.clear()
        _readers.clear()
