"""ctypes binding of librua_hip.so (the C ABI declared in include/rua.h).

The product path has no CPU fallback: if the shared library is missing, or an op is handed a
tensor that does not live on a HIP device, we raise.  PyTorch is used only for device memory and
streams; every kernel is launched through the plain-pointer C ABI on torch's current stream.
"""
import ctypes
import os
import threading
from ctypes import POINTER, Structure, c_char_p, c_int, c_int32, c_int64, c_uint64, c_void_p

import torch  # noqa: F401  (must be imported first: maps the HIP runtime our library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RUA_LIB_PATH') or os.path.join(_HERE, 'librua_hip.so')   # env: developer A/B of builds

ABI_VERSION = 6
# enum rua_kind
CAT, LEFT, PACK, RIGHT, LIST = 0, 1, 2, 3, 4
# enum rua_tmap
T_SHIFT, T_ROLL, T_REV_S, T_REV_D, T_ZERO = 0, 1, 2, 3, 4
TIES_FINAL = 2      # rua_segment_reduce_backward include_self: `ties` came complete from the forward
BWD_FILL_PADDING = 0x100   # ... OR-ed in: the call itself zeroes the padding rows of a padded layout
BWD_TIES_POSITIVE = 0x200  # ... OR-ed in (max/min): torch.segment_reduce's tie rule — ties share g only where g > 0
MOVE_SCATTER = 1
MOVE_NT_ON, MOVE_NT_OFF = 2, 4       # rua.h: force / forbid non-temporal payload accesses
MOVE_NO_NARROW = 2048               # rua.h: developer A/B — rows of one vector through the generic kernel
OP_SCRATCH_CLEAN, OP_NO_EMPTY = 0x100, 0x200      # rua.h: bits OR-ed into `op` (persistent zeroed extreme scratch; no sequence is empty: proven)
EXTREME_WORDS = 1027                               # rua.h: RUA_EXTREME_WORDS
OP_SHORT_SEQS = 0x400      # rua.h: a CattedSequence of short sequences, none far above the average (a hint)
# enum rua_dtype / rua_op
F32, BF16, F16, F64 = 0, 1, 2, 3
SUM, MEAN, MAX, MIN, PROD, LOGSUMEXP = 0, 1, 2, 3, 4, 5

I64, I32, I16, I8, U8 = 4, 5, 6, 7, 8

DTYPES = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16, torch.float64: F64}
# integer element types: rua_segment_reduce over a CAT layout only (scatter_* on integer tensors, reduce.py:6-23)
INT_DTYPES = {torch.int64: I64, torch.int32: I32, torch.int16: I16, torch.int8: I8, torch.uint8: U8}


class RuaLayout(Structure):
    """struct rua_layout (include/rua.h)."""
    _fields_ = [
        ('kind', c_int32), ('tile_t_log2', c_int32),
        ('n_rows', c_int64), ('B', c_int64), ('T_phys', c_int64), ('T_log', c_int64),
        ('lens', c_void_p), ('len_add', c_int64), ('off', c_void_p),
        ('boff', c_void_p), ('T', c_int64), ('sorted', c_void_p), ('unsorted', c_void_p),
        ('bptr', c_void_p), ('tptr', c_void_p),
        ('bsz', c_void_p), ('tile_start', c_void_p), ('n_tchunks', c_int64), ('n_tiles', c_int64),
    ]


class RuaError(RuntimeError):
    pass


# every symbol include/rua.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'rua_scan_ws_elems': (c_int64, [c_int64]),
    'rua_exclusive_scan_i64': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'rua_pack_meta': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'rua_pack_prepare': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    'rua_lens_from_pack': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    'rua_enum_rows': (c_int, [POINTER(RuaLayout), c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rua_mask': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int32, c_uint64, c_uint64, c_void_p]),
    'rua_move_rows': (c_int, [POINTER(RuaLayout), POINTER(RuaLayout), c_int32, c_int64, c_void_p, c_void_p,
                              c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    'rua_reduce_ws_bytes': (c_int64, [c_int64, c_int64, c_int32, c_int64]),
    'rua_reduce_team_waves': (c_int, [c_int64, c_int64, c_int64]),
    'rua_segment_reduce': (c_int, [POINTER(RuaLayout), c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                   c_int32, c_uint64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'rua_pack_reduce': (c_int, [POINTER(RuaLayout), POINTER(RuaLayout), c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                c_int32, c_uint64, c_void_p, c_int64, c_void_p, c_void_p]),
    'rua_segment_reduce_backward': (c_int, [POINTER(RuaLayout), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int64, c_int32, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_void_p]),
    'rua_scatter_self_grad': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int32, c_int32, c_int32, c_void_p]),
    'rua_fill_empty': (c_int, [POINTER(RuaLayout), c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    'rua_bucket_ws_elems': (c_int64, [c_int64, c_int64]),
    'rua_index_buckets': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rua_host_sort_desc': (c_int, [c_void_p, c_int64, c_void_p, c_int32]),
    'rua_host_sort_desc_begin': (c_int, [c_void_p, c_int64, c_void_p, c_int32]),
    'rua_host_sort_desc_end': (c_int, []),
    'rua_host_sort_heap_segments': (c_int64, []),
    'rua_host_batch_sizes': (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    'rua_host_pack_scans': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    'rua_abi_version': (c_int, []),
    'rua_build_target': (c_char_p, []),
}

_lib = None


def load():
    """Load librua_hip.so (once).  Raises RuaError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuaError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C torchrua_amd/csrc`. torchrua_amd has no CPU/eager fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table disagree
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rua_abi_version() != ABI_VERSION:
        raise RuaError(f'ABI mismatch: library reports {lib.rua_abi_version()}, binding expects {ABI_VERSION}')
    _lib = lib
    return lib


_tls = threading.local()       # .prev: the device stream_ptr() switched away from for the launch in flight


def check(code: int, what: str) -> None:
    prev = getattr(_tls, 'prev', None)
    if prev is not None:           # stream_ptr() made another device current for this one launch: hand it back
        _tls.prev = None
        torch.cuda.set_device(prev)
    if code == 0:
        return
    if code < 0:
        names = {-1: 'RUA_EINVAL', -2: 'RUA_EALIGN', -3: 'RUA_ERANGE'}
        raise RuaError(f'{what}: rejected argument ({names.get(code, code)})')
    raise RuaError(f'{what}: HIP error {code}')


def stream_ptr(device) -> int:
    """torch's current stream on `device` — the device the TENSORS live on, whatever the thread's current device is,
    as with any torch op (and therefore with the reference: core/get.py:25-29 indexes on `self.data.device`).  A HIP
    launch goes to the calling thread's current device, so when that is another one it is switched for this ONE launch:
    every call site is `check(lib.rua_xxx(..., stream_ptr(dev)), name)`, the stream argument is evaluated last before
    the call and check() runs right after it and switches back.  (One process per GPU never takes this branch.)"""
    cur = torch.cuda.current_device()
    idx = cur if device.index is None else device.index
    if idx != cur:
        if getattr(_tls, 'prev', None) is None:
            _tls.prev = cur
        torch.cuda.set_device(idx)
    # the raw hipStream_t of torch's current stream; torch.cuda.current_stream() builds a Stream object (~6 us a call)
    return torch._C._cuda_getCurrentRawStream(idx)


def require_device(*tensors) -> torch.device:
    """All tensors must sit on one HIP device; there is no host path."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuaError(
                'torchrua_amd runs on MI355X only: got a tensor on '
                f'{t.device}; move it to a HIP device (there is no CPU fallback by design)')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuaError(f'tensors on different devices: {dev} vs {t.device}')
    return dev


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
