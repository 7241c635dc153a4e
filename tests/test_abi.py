"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/rua.h
declares, the Python surface carries the reference's names, and nothing computes off-GPU."""
import ctypes
import os
import re

import pytest
import torch

import torchrua_amd as ta
from torchrua_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, 'include', 'rua.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rua_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/rua.h but not exported'
    assert sorted(_lib.SYMBOLS) == names, 'ctypes table and header disagree'
    loaded = _lib.load()
    assert loaded.rua_abi_version() == _lib.ABI_VERSION == 6
    assert loaded.rua_build_target() == b'gfx950'
    assert loaded.rua_scan_ws_elems(1) >= 1 and loaded.rua_scan_ws_elems(1 << 20) >= 512


def test_layout_struct_matches_header_size():
    # 19 fields, all 8-byte aligned after the two int32s
    assert ctypes.sizeof(_lib.RuaLayout) == 8 + 17 * 8


def test_reference_surface_is_present():
    """symbols a replacement must export."""
    for n in ('C L P R T Z CattedSequence LeftAlignedSequence RightAlignedSequence PackedSequence '
              'major_sizes_to_ptr get_offsets invert_permutation get_mask compose '
              'segment_max segment_min segment_sum segment_mean segment_prod segment_logsumexp segment_head '
              'segment_last scatter_max scatter_min scatter_sum scatter_mean scatter_prod scatter_logsumexp').split():
        assert hasattr(ta, n), n
    methods = ('new cat left pack right cat_view left_view pack_view right_view size ptr idx offsets raw '
               '__getitem__ __setitem__ head last roll rev trunc seg mask bmask fmask split tolist').split()
    for cls in (ta.C, ta.L, ta.P, ta.R):
        for m in methods:
            assert hasattr(cls, m), f'{cls.__name__}.{m}'
    for cls in (ta.C, ta.L, ta.R):
        for m in 'to double float half long int short char byte cpu cuda detach'.split():
            assert hasattr(cls, m), f'{cls.__name__}.{m}'
    assert ta.C._fields == ('data', 'token_sizes') == ta.L._fields == ta.R._fields
    assert ta.P is torch.nn.utils.rnn.PackedSequence


def test_no_cpu_fallback():
    """The product path fails loudly off-GPU instead of computing on the host."""
    seqs = [torch.randn(3, 2), torch.randn(2, 2)]
    c = ta.C.new(seqs)                      # construction is plain torch.cat
    assert c.data.shape == (5, 2) and c.token_sizes.tolist() == [3, 2]
    for call in (c.pack, c.left, c.right, lambda: c.roll(1), c.last, lambda: c.head(1), c.ptr,
                 lambda: ta.segment_sum(c.data, c.token_sizes),
                 lambda: ta.scatter_sum(torch.zeros(2, 2), torch.tensor([0, 1, 1, 0, 0]), c.data),
                 lambda: ta.get_offsets(c.token_sizes)):
        with pytest.raises(ta.RuaError):
            call()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'torchrua_amd')
    for name in os.listdir(pkg):
        if name.endswith('.py'):
            text = open(os.path.join(pkg, name)).read()
            assert 'oracle' not in text.replace('the oracle', ''), f'{name} mentions the oracle'


def test_host_metadata_helpers():
    from torchrua_amd import _meta as M
    lens = torch.tensor([2, 4, 1, 3])
    assert M.batch_sizes_from_host_lens(lens, 4).tolist() == [4, 3, 2, 1]
    assert M.batch_sizes_from_host_lens(torch.tensor([3, 3]), 3).tolist() == [2, 2, 2]
    t = torch.zeros(3)
    M._memo_put(t, 'k', 7)
    assert M._memo_get(t, 'k') == 7
    t.add_(1)                                # in-place edit invalidates what was derived from it
    assert M._memo_get(t, 'k') is None


def test_reducer_split_policy():
    """Host-side choice of the reducer's part size (no GPU involved)."""
    from torchrua_amd import _meta as M

    class FakeLay:
        def __init__(self, n_rows, B, max_len):
            self.n_rows, self.B, self.max_len = n_rows, B, max_len

    ns = FakeLay(17_046_960, 65536, 512)               # north-star shape: nothing to split
    assert M.reduce_split_rows(ns, 1024) == 0
    assert M.reduce_split_rows(FakeLay(34_000_000, 65536, 1024), 2048) == 0    # cfg4
    assert M.reduce_split_rows(FakeLay(1_070_000, 4096, 512), 512) == 0        # cfg2: LPT hides 512-row sequences
    # few units: a team of 4 waves per unit (rows up to 1 KiB on the vector path) fills the chip without the three
    # launches of the split — rows of 8 (mod 16) bytes included since round 3 (16-byte lanes with an overlapping last
    # lane); rows the team kernel does not take (wider than 1 KiB, widths that are no multiple of 8) are still cut
    assert M.reduce_split_rows(FakeLay(133_000, 512, 512), 1024) == 0
    assert M.reduce_split_rows(FakeLay(133_000, 512, 512), 2048) == 64
    assert M.reduce_split_rows(FakeLay(133_000, 512, 512), 1000) == 0
    assert M.reduce_split_rows(FakeLay(133_000, 512, 512), 1004) == 65
    assert M.reduce_split_rows(FakeLay(533_000, 2048, 512), 128) == 0          # narrow rows: the longest walk is 16 us
    assert 64 <= M.reduce_split_rows(FakeLay(1_131_008, 2048, 1_000_000), 1024) <= 256   # one giant sequence
    assert M.reduce_split_rows(FakeLay(100_000_000, 64, 5_000_000), 1024) == 4096
    assert M.reduce_split_rows(FakeLay(536_149, 16384, None), 1024) == 0       # cfg3, device-only lengths
    assert M.reduce_split_rows(FakeLay(5_000_000, 100, None), 1024) == 610     # few long sequences, lengths unknown
    blind = M.reduce_split_rows(FakeLay(1_070_000, 4096, None), 512)           # cfg2 with device-only lengths:
    assert blind > 512                                                         # armed, but nothing of cfg2 is cut
    assert M.reduce_split_rows(FakeLay(200, 3, 150), 1024) == 0
    # the buckets of a scatter_* (sizes on the device, one hot bucket is ordinary): always armed from 64 MB of payload
    hot = FakeLay(4_000_000, 100_000, None)
    assert M.reduce_split_rows(hot, 128) == 0
    assert M.reduce_split_rows(FakeLay(17_046_960, 100_000, None), 128) > 0      # 2 GB with device-only lengths: armed too
    hot.heavy_tail = True
    assert M.reduce_split_rows(hot, 128) > 0 and M.reduce_split_rows(hot, 1024, team_ok=False) > 0
    small = FakeLay(20_000, 5_000, None)
    small.heavy_tail = True
    assert M.reduce_split_rows(small, 128) == 0
    # RUA_OP_SHORT_SEQS (adjacent sequences of a CattedSequence side by side in a wave): host-known lengths only, short
    # on average by row width, nothing far above the average
    from torchrua_amd import _ops as O

    class CatLay(FakeLay):
        kind = _lib.CAT
    assert O.short_seqs_hint(CatLay(8_000_000, 500_000, 31), 32) == _lib.OP_SHORT_SEQS      # 16 rows on average, 32-byte rows
    assert O.short_seqs_hint(CatLay(8_000_000, 500_000, None), 32) == 0                     # lengths on the device only
    assert O.short_seqs_hint(CatLay(8_000_000, 500_000, 10_000), 32) == 0                   # one long sequence among them
    assert O.short_seqs_hint(CatLay(8_000_000, 500_000, 31), 1024) == 0                     # rows of a whole wave instruction
    assert O.short_seqs_hint(CatLay(17_046_960, 65536, 512), 32) == _lib.OP_SHORT_SEQS      # 260 on average, at most 512
    assert O.short_seqs_hint(CatLay(17_046_960, 65536, 5000), 32) == 0                      # ... with an outlier
    pk = CatLay(8_000_000, 500_000, 31)
    pk.kind = _lib.PACK
    assert O.short_seqs_hint(pk, 32) == 0
    # the wave-team rule the planner prices with IS the launcher's (rua_reduce_team_waves): a few fixed points
    lib = _lib.load()
    assert lib.rua_reduce_team_waves(133_000, 512, 1024) == 4 and lib.rua_reduce_team_waves(133_000, 512, 1000) == 4
    assert lib.rua_reduce_team_waves(133_000, 512, 1004) == 1 and lib.rua_reduce_team_waves(133_000, 512, 2048) == 1
    assert lib.rua_reduce_team_waves(17_046_960, 65536, 1024) == 1                 # plenty of units: one wave each
    assert lib.rua_reduce_team_waves(40_000, 512, 1024) == 2 and lib.rua_reduce_team_waves(8, 512, 16) == 1
    t = torch.tensor([3, 9, 2])
    assert M.known_max_len(t) == 9 and M.known_max_len(None) is None


def test_dtype_and_device_movers():
    """layout/cat.py:13-59 and twins: to / double / float / half / long / int / short / char / byte / cpu / detach return
    the same container type with the payload converted and the lengths left as they are (plain torch plumbing: runs on
    the CPU)."""
    import torchrua_amd as ta
    for cls, shape in ((ta.C, (6, 3)), (ta.L, (2, 4, 3)), (ta.R, (2, 4, 3))):
        z = cls(torch.randn(shape, requires_grad=True), torch.tensor([2, 4]))
        for name, dtype in (('double', torch.double), ('float', torch.float), ('half', torch.half), ('long', torch.long),
                            ('int', torch.int), ('short', torch.short), ('char', torch.int8), ('byte', torch.uint8)):
            out = getattr(z, name)()
            assert type(out) is cls and out.data.dtype == dtype and out.token_sizes.dtype == torch.long
            assert torch.equal(out.token_sizes, z.token_sizes)
        out = z.to(dtype=torch.bfloat16, device=torch.device('cpu'))
        assert type(out) is cls and out.data.dtype == torch.bfloat16 and out.data.device.type == 'cpu'
        assert type(z.cpu()) is cls and not z.detach().data.requires_grad and z.detach().data.data_ptr() == z.data.data_ptr()


def test_launches_follow_the_tensors_device(monkeypatch):
    """torch (and so the reference) runs an op on the device its tensors live on, whatever the thread's
    current device; a HIP launch goes to the current device, so `_lib.stream_ptr` makes the tensors' device current
    for the ONE launch and `_lib.check` — which wraps every launch — hands the old one back, also when the launch fails."""
    import pytest
    import torch
    from torchrua_amd import _lib
    state, calls = {'cur': 0}, []
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: state['cur'])
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: (calls.append(d), state.__setitem__('cur', d)))
    monkeypatch.setattr(torch._C, '_cuda_getCurrentRawStream', lambda idx: 1000 + idx)
    assert _lib.stream_ptr(torch.device('cuda', 1)) == 1001 and state['cur'] == 1 and calls == [1]
    _lib.check(0, 'launch')
    assert state['cur'] == 0 and calls == [1, 0]
    assert _lib.stream_ptr(torch.device('cuda', 0)) == 1000 and _lib.stream_ptr(torch.device('cuda')) == 1000
    _lib.check(0, 'launch')
    assert calls == [1, 0]                                   # the tensors' device was current already: nothing switched
    _lib.stream_ptr(torch.device('cuda', 1))
    with pytest.raises(_lib.RuaError):
        _lib.check(-1, 'launch')
    assert state['cur'] == 0
