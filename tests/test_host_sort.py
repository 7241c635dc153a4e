"""The host side of pack(): rua_host_sort_desc must return EXACTLY torch.sort(lens, descending=True)[1] — the
reference's call (core/view.py:48), tie order included — for any input and any thread count, and
rua_host_batch_sizes the reference's get_mask(..).sum(0) (core/view.py:55).  No GPU involved."""
import os

import numpy as np
import pytest
import torch

from torchrua_amd import _lib as L
from torchrua_amd import _meta as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ours(keys: torch.Tensor, threads: int) -> torch.Tensor:
    out = torch.empty_like(keys)
    assert L.load().rua_host_sort_desc(keys.data_ptr(), keys.numel(), out.data_ptr(), threads) == 0
    return out


def _ref(keys: torch.Tensor) -> torch.Tensor:
    return torch.sort(keys, descending=True)[1]


@pytest.mark.parametrize('threads', [1, 2, 5, 8])
def test_same_permutation_as_torch_sort_random(threads):
    g = torch.Generator().manual_seed(threads)
    for trial in range(120):
        n = int(torch.randint(1, 30000, (1,), generator=g))
        hi = [1, 4, 16, 512, 100000, 1 << 50][trial % 6]
        keys = torch.randint(0, hi + 1, (n,), generator=g)
        assert torch.equal(_ours(keys, threads), _ref(keys)), (n, hi)


def test_same_permutation_on_every_small_size():
    g = torch.Generator().manual_seed(0)
    for n in range(0, 70):                       # around the 16-element leaf and the first partitions
        for hi in (0, 1, 2, 5, 1000):
            keys = torch.randint(0, hi + 1, (n,), generator=g)
            assert torch.equal(_ours(keys, 1), _ref(keys)), (n, hi)


@pytest.mark.parametrize('n', [17, 1000, 65536])
def test_same_permutation_on_structured_inputs(n):
    up = torch.arange(n)
    cases = {'ascending': up, 'descending': up.flip(0), 'constant': torch.full((n,), 7),
             'organ pipe': torch.cat([up[:n // 2], up[:n - n // 2].flip(0)]),
             'periodic': (up * 7919) % 13, 'two values': (up >= n // 2).long(), 'sawtooth': up % 100,
             'negative': -up % 17 - 5}
    for name, keys in cases.items():
        for threads in (1, 4):
            assert torch.equal(_ours(keys.contiguous(), threads), _ref(keys)), name


def test_baseline_shapes():
    for seed, B, lo, hi in ((2, 4096, 8, 512), (3, 16384, 1, 64), (4, 65536, 16, 1024), (5, 65536, 8, 512),
                            (6, 524288, 8, 512)):
        keys = torch.randint(lo, hi + 1, (B,), generator=torch.Generator().manual_seed(seed))
        ref = _ref(keys)
        for threads in (1, 8):
            assert torch.equal(_ours(keys, threads), ref), (B, threads)


def test_depth_budget_and_heap_branch():
    """Inputs built by an adversary (scripts/exp/sort_killer.cpp) that exhaust the introsort's depth budget, with and
    without ties: the heap-sort branch must be taken AND still agree with torch.sort."""
    lib = L.load()
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'sort_killer.npz'))
    for name in z.files:
        keys = torch.from_numpy(z[name].astype(np.int64))
        before = lib.rua_host_sort_heap_segments()
        got = _ours(keys, 1)
        assert lib.rua_host_sort_heap_segments() > before, f'{name}: never left the quicksort levels'
        assert torch.equal(got, _ref(keys)), name
        assert torch.equal(_ours(keys, 4), _ref(keys)), name


def test_argument_checks():
    lib = L.load()
    assert lib.rua_host_sort_desc(None, 0, None, 1) == 0
    assert lib.rua_host_sort_desc(None, 4, None, 1) < 0
    assert lib.rua_host_sort_desc(None, -1, None, 1) < 0
    assert lib.rua_host_batch_sizes(None, 0, 0, None) == 0
    bad = torch.tensor([3, -1])
    out = torch.empty(3, dtype=torch.long)
    assert lib.rua_host_batch_sizes(bad.data_ptr(), 2, 3, out.data_ptr()) < 0


def test_batch_sizes_from_host_lens():
    g = torch.Generator().manual_seed(9)
    for B, hi in ((1, 5), (7, 3), (1000, 64), (4096, 512)):
        lens = torch.randint(0, hi + 1, (B,), generator=g)
        T = int(lens.max())
        ref = (lens[:, None] > torch.arange(T)[None, :]).sum(0)            # core/view.py:55 on the B x T mask
        assert torch.equal(M.batch_sizes_from_host_lens(lens, T), ref)
    assert M.batch_sizes_from_host_lens(torch.zeros(3, dtype=torch.long), 0).numel() == 0


def test_host_sort_policy(monkeypatch):
    """The Python layer: self-test passes here, so the library sort is in use; RUA_HOST_SORT=torch keeps the
    reference's own call; both return the same permutation into a caller-provided buffer."""
    keys = torch.randint(0, 9, (5000,), generator=torch.Generator().manual_seed(1))
    ref = _ref(keys)
    monkeypatch.setattr(M, '_host_sort_threads', None)
    monkeypatch.delenv('RUA_HOST_SORT', raising=False)
    assert torch.equal(M.host_sort_desc(keys), ref) and M._host_sort_threads >= 1
    buf = torch.empty(5000, dtype=torch.long)
    assert M.host_sort_desc(keys, out=buf) is buf and torch.equal(buf, ref)
    monkeypatch.setattr(M, '_host_sort_threads', None)
    monkeypatch.setenv('RUA_HOST_SORT', 'torch')
    assert torch.equal(M.host_sort_desc(keys), ref) and M._host_sort_threads == 0
    assert torch.equal(M.host_sort_desc(keys, out=buf), ref)
    monkeypatch.setattr(M, '_host_sort_threads', None)
    monkeypatch.setenv('RUA_HOST_SORT', '3')
    assert torch.equal(M.host_sort_desc(keys), ref) and M._host_sort_threads == 3


@pytest.mark.parametrize('sanitizer', ['thread', 'address,undefined'])
def test_host_sort_under_sanitizers(tmp_path, sanitizer):
    """The host sort keeps worker threads and a grow-only scratch between calls: build it with the sanitizers (CPU
    build only — the GPU pool has none) and hammer it from three threads at once (tests/c/host_sort_stress.cpp)."""
    import shutil
    import subprocess
    if not shutil.which('g++'):
        pytest.skip('no g++')
    exe = str(tmp_path / 'stress')
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', f'-fsanitize={sanitizer}', '-pthread',
                    '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'c', 'host_sort_stress.cpp'),
                    os.path.join(ROOT, 'torchrua_amd', 'csrc', 'rua_host.cpp'), '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    if 'unexpected memory mapping' in out.stderr:
        pytest.skip('this kernel\'s address-space layout is one the sanitizer runtime cannot run under')
    assert out.returncode == 0 and 'mismatches 0' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'WARNING: ThreadSanitizer' not in out.stderr and 'ERROR: AddressSanitizer' not in out.stderr \
        and 'runtime error' not in out.stderr, out.stderr[-4000:]


def test_async_sort_is_the_same_sort():
    """rua_host_sort_desc_begin / _end (the sort on the library's helper thread, so that pack() with device-only lengths
    can do the rest of its host work meanwhile): the same permutation as the synchronous call and as torch.sort, one job
    at a time, and rua_host_pack_scans next to it."""
    from torchrua_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    assert lib.rua_host_sort_desc_end() == -1                              # no job posted
    for n, hi in ((1, 5), (17, 2), (5000, 3), (65536, 512), (100000, 1 << 40)):
        keys = torch.randint(0, hi + 1, (n,), generator=g)
        out = torch.empty_like(keys)
        assert lib.rua_host_sort_desc_begin(keys.data_ptr(), n, out.data_ptr(), 4) == 0
        assert lib.rua_host_sort_desc_begin(keys.data_ptr(), n, out.data_ptr(), 4) == -1      # busy
        # the caller's share of the interval: batch_sizes and the two scans
        T = int(keys.max()) if hi < 100000 else 0
        bsz = torch.empty(T, dtype=torch.long)
        boff, off = torch.empty(T, dtype=torch.long), torch.empty(n, dtype=torch.long)
        if T:
            assert lib.rua_host_batch_sizes(keys.data_ptr(), n, T, bsz.data_ptr()) == 0
        assert lib.rua_host_pack_scans(keys.data_ptr(), n, bsz.data_ptr() if T else None, T, boff.data_ptr() if T else None,
                                       off.data_ptr()) == 0
        assert lib.rua_host_sort_desc_end() == 0
        assert torch.equal(out, torch.sort(keys, descending=True)[1]), n
        assert torch.equal(off, torch.cumsum(keys, 0) - keys)
        if T:
            assert torch.equal(bsz, (keys[None, :] > torch.arange(T)[:, None]).sum(1))
            assert torch.equal(boff, torch.cumsum(bsz, 0) - bsz)
    assert lib.rua_host_sort_desc_end() == -1
