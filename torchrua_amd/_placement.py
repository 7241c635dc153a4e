"""Where a LARGE output of the row mover is put.

On the MI355X the time of a multi-GB move depends on WHICH allocations it runs between: the buffers of a process fall
into a few classes (profiles/r02_placement.txt; they look like the ranks of the HBM3E stacks, but physical addresses
are not visible from user space), and a move whose source and destination sit in the same class is ~6 % slower than
one between classes — the same kernel, the same bytes (north-star pack: 6.05 vs 5.65 ms).  torch's caching allocator
hands out whichever cached block fits, so a pipeline that alternates between two output blocks runs at the mix.

This module lets the mover learn which blocks are good for which source, from its own launches:

  * every launch into a large output (>= MIN_BYTES) is bracketed by two events on its stream; finished pairs are read
    back later, without ever waiting, and the time is filed under (what moved, source block, output block) — blocks
    are known by their storage address;
  * when torch offers a block whose time for this source is known to be worse than the best one seen by more than
    TOLERANCE, the offer is held and another block is asked for (a held block is alive, so the allocator has to
    come up with a different one — only cached blocks are asked for here); the rejected offers go straight back to
    the caching allocator;
  * a block nobody has timed yet is simply used, and until EXPLORE different blocks have each been moved into
    MIN_SAMPLES times for a source, an offer whose cost is already known is held in favour of one whose cost is not
    (two equally slow blocks would otherwise look fine for ever) — a cached one, or, while the card has ample room,
    a fresh one: that is the exploration, and it is what may grow the cache by EXPLORE - 2 blocks.

A caching allocator recycles a handful of blocks, so after the first steps every pair is known and the choice is a
dictionary lookup plus, at worst, a couple of cached alloc / free pairs per call.  No synchronisation, no extra GPU
work; the price is up to EXPLORE - 2 more cached blocks of the output's size, some of which end up parked in the
allocator's cache because they are slow for this source.

RUA_PLACEMENT=0 switches it off; RUA_PLACEMENT_MIN_BYTES moves the threshold (default 2 GiB).  Nothing here changes
a result: only which block the output lives in.
"""
import os
import threading
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

ENABLED = os.environ.get('RUA_PLACEMENT', '1') != '0'
MIN_BYTES = int(os.environ.get('RUA_PLACEMENT_MIN_BYTES', 2 << 30))
TOLERANCE = 0.015          # a block within this of the best cost for its source is taken as it comes (launches between
                           # one pair of blocks repeat to ~0.3 %)
EXPLORE = int(os.environ.get('RUA_PLACEMENT_EXPLORE', 4))   # distinct blocks to try for a source before trusting the best seen
MAX_HELD = EXPLORE + 1     # offers in hand at most while choosing: enough to see every block that was tried, so that
                           # settling for the best one on offer really is the best one free
MIN_SAMPLES = 2            # a block is only called slow after this many launches (the first one into fresh memory
                           # also pays for its page tables)
PENDING_MAX = 256
TABLE_MAX = 4096           # (source, output) pairs remembered; beyond that everything is learned afresh
_lock = threading.Lock()

Key = Tuple[str, int, int, int]                      # (move name, output bytes, source bytes, source storage)
_times: Dict[Tuple[Key, int], Tuple[float, int]] = {}   # (key, output storage) -> (fastest ms, launches timed)
_pending: List[Tuple[Key, int, torch.cuda.Event, torch.cuda.Event]] = []
_tried: Dict[Key, set] = {}                         # output blocks launched into, per key (timed or still pending)
_blocks: Dict[Tuple[int, int], set] = {}            # (device, bytes) -> storage addresses met for outputs of this size
_frozen: Dict[Tuple[int, int], bool] = {}           # (device, bytes) -> holding switched off (see _may_ask_again)
_launched: Dict[Tuple[Key, int], int] = {}          # (key, output storage) -> launches into it so far
_no_room: Dict[Tuple[int, int], bool] = {}          # (device, bytes) -> the card had no room for a fresh block
_readers: Dict[Tuple[str, int], Dict[int, float]] = {}   # (reader kernel, bytes) -> {block: fastest ms}: what the NEXT
                                                    # kernel pays to read a block this module placed
stats = {'timed': 0, 'explored': 0, 'rejected': 0, 'taken_good': 0, 'taken_untimed': 0, 'settled': 0}


def _base(t: Tensor) -> int:
    return t.untyped_storage().data_ptr()


def _harvest() -> None:
    """File the launches that have finished; never waits."""
    if len(_times) > TABLE_MAX:
        _times.clear()
        _tried.clear()
        _blocks.clear()
        _readers.clear()
        _launched.clear()
    keep = []
    for item in list(_pending):
        key, ob, e0, e1 = item
        if not e1.query():
            keep.append(item)
            continue
        ms = e0.elapsed_time(e1)
        if key[0].startswith('read:'):
            table = _readers.setdefault((key[0], key[1]), {})
            table[ob] = min(ms, table.get(ob, float('inf')))
            stats['timed'] += 1
            continue
        old = _times.get((key, ob))
        _times[(key, ob)] = (ms, 1) if old is None else (min(ms, old[0]), old[1] + 1)
        stats['timed'] += 1
    _pending[:] = keep[-PENDING_MAX:]


def _read_penalty(ob: int, nbytes: int) -> float:
    """ms a block costs its readers over the best block of its size, as far as readers have been timed on both."""
    worst = 0.0
    for (_name, n), table in _readers.items():
        if n == nbytes and ob in table and len(table) > 1:
            worst = max(worst, table[ob] - min(table.values()))
    return worst


def _verdict(key: Key, ob: int) -> Optional[float]:
    """The block's cost for this source — its move time plus what its readers pay — when it is KNOWN to be worse than
    the best block tried by more than TOLERANCE, else None (good, or not timed enough)."""
    hit = _times.get((key, ob))
    if hit is None or hit[1] < MIN_SAMPLES:
        return None
    best = float('inf')
    for b in _tried.get(key, ()):
        t = _times.get((key, b))
        if t is not None and (t[1] >= MIN_SAMPLES or b == ob):
            best = min(best, t[0] + _read_penalty(b, key[1]))
    cost = hit[0] + _read_penalty(ob, key[1])
    return cost if cost > best * (1.0 + TOLERANCE) else None


def _may_ask_again(key: Key, n_held: int, dev: torch.device, exploring: bool) -> bool:
    """May another block be asked for while `n_held` offers are held?  The allocator is not asked what it has cached
    (torch.cuda.memory_stats / hipMemGetInfo turned out to wait for the device: 1.6 ms of GPU idle per step when they
    sat on the steady path); the module keeps its own list of the blocks of this size it has met.  Outside exploration
    another block is asked for only when that list is long enough to leave one after the held offers and one output
    still alive from the previous call; should the allocator answer with a block never met before (it had nothing
    cached and grew), holding is switched off for this size — growth by one block, once."""
    size = (dev.index, key[1])
    if exploring:
        if len(_blocks.get(size, ())) >= n_held + 2:
            return True                              # probably cached
        if len(_tried.get(key, ())) >= EXPLORE or _no_room.get(size, False):
            return False
        free, _total = torch.cuda.mem_get_info(dev)  # (waits for the device: at most EXPLORE times per source,
        if free < 3 * key[1]:                        #  and never again for this size once the answer was no)
            _no_room[size] = True
            return False
        return True
    return not _frozen.get(size, False) and len(_blocks.get(size, ())) >= n_held + 2


def _met(key: Key, t: Tensor, dev: torch.device, exploring: bool) -> None:
    size = (dev.index, key[1])
    known = _blocks.setdefault(size, set())
    if _base(t) not in known:
        if known and not exploring and len(known) >= 2:
            _frozen[size] = True                     # the allocator grew to serve a held offer: stop holding
        known.add(_base(t))


def key_for(name: str, out_bytes: int, src: Tensor) -> Optional[Key]:
    if not ENABLED or out_bytes < MIN_BYTES or torch.cuda.is_current_stream_capturing():
        return None
    return (name, out_bytes, src.numel() * src.element_size(), _base(src))


def empty_for(shape, dtype: torch.dtype, dev: torch.device, key: Key) -> Tensor:
    """torch.empty(shape) for an output the mover is about to fill — in a block that is not known to be slow for this
    source, if one can be had."""
    with _lock:
        return _empty_for(shape, dtype, dev, key)


def _empty_for(shape, dtype: torch.dtype, dev: torch.device, key: Key) -> Tensor:
    _harvest()
    out = torch.empty(shape, dtype=dtype, device=dev)
    tried = _tried.setdefault(key, set())

    def fresh(t: Tensor) -> bool:          # not moved into MIN_SAMPLES times yet: its cost is not known
        return _launched.get((key, _base(t)), 0) < MIN_SAMPLES

    exploring = len(tried) < EXPLORE or any(_launched.get((key, b), 0) < MIN_SAMPLES for b in tried)
    _met(key, out, dev, True)
    if exploring and not fresh(out):
        # exploration: prefer a block whose cost for this source is not known yet — one never moved into, or moved
        # into once (the first launch into fresh memory also pays for its page tables and does not count)
        spare = [out]
        while len(spare) < MAX_HELD and _may_ask_again(key, len(spare), dev, True):
            nxt = torch.empty(shape, dtype=dtype, device=dev)
            _met(key, nxt, dev, True)
            if fresh(nxt):
                stats['explored'] += 1
                return nxt
            spare.append(nxt)
        out = spare[0]
        del spare
    slow = _verdict(key, _base(out))
    if slow is None:
        stats['taken_good' if (key, _base(out)) in _times else 'taken_untimed'] += 1
        return out
    held: List[Tuple[float, Tensor]] = [(slow, out)]
    while len(held) < MAX_HELD and _may_ask_again(key, len(held), dev, False):
        out = torch.empty(shape, dtype=dtype, device=dev)      # the held offers are alive: this is another block
        _met(key, out, dev, False)
        slow = _verdict(key, _base(out))
        if slow is None:
            stats['rejected'] += len(held)
            stats['taken_good' if (key, _base(out)) in _times else 'taken_untimed'] += 1
            return out
        held.append((slow, out))
    held.sort(key=lambda x: x[0])
    stats['rejected'] += len(held) - 1
    stats['settled'] += 1
    return held[0][1]


def begin(key: Key, out: Tensor, stream: torch.cuda.Stream):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    with _lock:
        _tried.setdefault(key, set()).add(_base(out))
        _launched[(key, _base(out))] = _launched.get((key, _base(out)), 0) + 1
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
    """Drop everything learned (after torch.cuda.empty_cache(): a storage address may come back on other memory)."""
    _times.clear()
    _pending.clear()
    _tried.clear()
    _blocks.clear()
    _frozen.clear()
    _readers.clear()
    _launched.clear()
    _no_room.clear()

