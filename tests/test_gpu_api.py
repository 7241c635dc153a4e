"""API-surface behaviours next to the kernels: dtype/device movers, *_view constructors, the opt-in Tensor
indexing patch, the torchrua alias, PackedSequences made by torch itself."""
import sys

import pytest
import torch
from torch.nn.utils.rnn import pack_sequence, pad_sequence

import torchrua_amd as ta
from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _xs(lens, dim=5):
    return [torch.randn(n, dim, device=DEV) for n in lens]


def test_movers_keep_structure():
    c = ta.C.new(_xs([3, 1, 4]))
    for z in (c, c.left(), c.right()):
        assert z.double().data.dtype == torch.float64 and z.half().data.dtype == torch.float16
        assert z.long().data.dtype == torch.long and z.byte().data.dtype == torch.uint8
        assert type(z.float()) is type(z) and z.float().token_sizes.dtype == torch.long
        back = z.cpu()
        assert back.data.device.type == 'cpu' and back.token_sizes.device.type == 'cpu'
        again = back.cuda()
        assert torch.equal(again.data, z.data) and again.data.is_cuda
        assert not z.detach().data.requires_grad
        assert torch.equal(z.to(dtype=torch.float64).cat().data, c.data.double())


def test_views_share_storage_and_metadata():
    xs = _xs([2, 5, 3])
    c = ta.C.new(xs)
    p = c.pack()
    l = c.left()
    assert p.cat_view().data is p.data and p.cat_view().token_sizes.tolist() == [2, 5, 3]
    assert c.cat_view() is c and l.left_view(0) is l and p.pack_view() is p
    lv = c.left_view(-1.0)
    assert lv.data.shape == (3, 5, 5) and bool((lv.data == -1).all()) and lv.token_sizes.tolist() == [2, 5, 3]
    rv = p.right_view(7, dtype=torch.long)
    assert rv.data.dtype == torch.long and bool((rv.data == 7).all())
    pv = l.pack_view()
    assert pv.data is l.data and torch.equal(pv.batch_sizes, p.batch_sizes)
    assert torch.equal(pv.sorted_indices, p.sorted_indices) and torch.equal(pv.unsorted_indices, p.unsorted_indices)
    assert c.raw() is c.data and l.raw().shape == (15, 5)
    assert c.size() == (3, 5, 5) == l.size() == p.size()


def test_z_keys_and_tensor_patch():
    xs = _xs([2, 4, 1])
    c = ta.C.new(xs)
    p = c.pack()
    idx = p.idx()                                   # a PackedSequence of row indices
    assert torch.equal(p[idx].data, p.data)         # container[Z] re-wraps
    assert torch.equal(c[c.idx().roll(1)].data, c.roll(1).data)
    plain = torch.arange(10, device=DEV)
    with pytest.raises((TypeError, IndexError, RuntimeError)):
        plain[idx]                                  # not patched by default
    ta.patch_tensor_indexing()
    try:
        got = p.data[idx]                           # reference core/get.py:11-18 behaviour, opt-in
        assert isinstance(got, ta.P) and torch.equal(got.data, p.data)
        buf = torch.zeros_like(p.data)
        buf[idx] = p.data
        assert torch.equal(buf, p.data)
        assert plain[2].item() == 2                 # ordinary indexing untouched
    finally:
        ta.unpatch_tensor_indexing()
    with pytest.raises((TypeError, IndexError, RuntimeError)):
        plain[idx]


def test_alias_module():
    """`install_as_torchrua()` = what `import torchrua` gives with the reference: the names AND the import-time
    patch of Tensor.__getitem__/__setitem__ (core/get.py:11-18, core/set.py:10-18)."""
    ta.install_as_torchrua()
    try:
        import torchrua
        assert torchrua is sys.modules['torchrua_amd'] and torchrua.C is ta.C
        assert torchrua.segment_sum is ta.segment_sum
        xs = _xs([3, 1, 4, 2])
        c = torchrua.C.new(xs)
        for z in (c, c.pack(), c.left(), c.right()):
            i = z.idx()                                         # rows of z.raw() in z's own order
            got = z.raw()[i]                                    # tensor[Z]: select/roll.py:21 spells roll this way
            assert type(got) is type(i) and torch.equal(got.data, z.raw()[i.data])
            if isinstance(z, (torchrua.C, torchrua.P)):
                assert torch.equal(got.data, z.data)
            buf = torch.zeros_like(z.raw())
            buf[i] = got.data                                   # tensor[Z] = value (core/set.py:10-18)
            ref = torch.zeros_like(z.raw())
            ref.index_put_((i.data,), got.data)
            assert torch.equal(buf, ref)
        # the reference's own spelling of roll for a padded layout goes through tensor[Z]
        l = c.left()
        rolled = l.raw()[l.idx().cat().roll(1).left()]          # select/roll.py:19-20 via the patched Tensor
        assert isinstance(rolled, torchrua.L) and torch.equal(rolled.data, l.roll(1).data)
        # gradients flow through tensor[Z]
        x = torch.randn(10, 5, device=DEV, requires_grad=True)
        k = c.idx().roll(1)
        x[k].data.sum().backward()
        assert torch.equal(x.grad, torch.ones_like(x))
        assert torch.arange(5, device=DEV)[2].item() == 2       # ordinary indexing untouched
    finally:
        ta.unpatch_tensor_indexing()


def test_packed_sequences_built_by_torch():
    """PackedSequence from torch.nn.utils.rnn (both enforce_sorted modes) feeds every consumer."""
    lens = [5, 4, 4, 2, 1]
    xs = _xs(lens)
    exp_pad = pad_sequence(xs, batch_first=True)
    exp_sum = torch.stack([x.sum(0) for x in xs])
    for enforce in (False, True):
        p = pack_sequence(xs, enforce_sorted=enforce)          # sorted_indices is None when enforce_sorted=True
        assert torch.equal(p.left().data, exp_pad)
        assert torch.equal(p.cat().data, torch.cat(xs))
        assert p.cat().token_sizes.tolist() == lens
        torch.testing.assert_close(ta.reduce_sum(p), exp_sum, rtol=1e-5, atol=1e-5)
        assert torch.equal(p.last(), torch.stack([x[-1] for x in xs]))
        assert torch.equal(p.roll(1).cat().data, torch.cat([x.roll(1, 0) for x in xs]))
        bp, tp = p.ptr()
        assert bp[:5].tolist() == [0, 1, 2, 3, 4] and tp[:5].tolist() == [0] * 5


def test_wrong_device_and_dtype_errors():
    c = ta.C.new(_xs([2, 3]))
    with pytest.raises(ta.RuaError):
        ta.segment_sum(c.data.long(), c.token_sizes)           # integer reductions are not in the reference either
    with pytest.raises(ta.RuaError):
        ta.scatter_sum(torch.zeros(2, 5, device=DEV), torch.tensor([0, 1, 1, 0, 0]), c.data)   # index on the CPU
    with pytest.raises(ta.RuaError):
        ta.scatter_max(torch.zeros(5, 2, device=DEV), torch.zeros(5, dtype=torch.long, device=DEV), c.data, dim=2)   # no such dim


def test_product_process_never_loads_the_oracle(tmp_path):
    """A process that only uses the product maps librua_hip.so and nothing from oracle/ (checked in /proc/self/maps)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, torch
sys.path.insert(0, %r)
import torchrua_amd as ta
xs = [torch.randn(n, 8, device="cuda:0", requires_grad=True) for n in (3, 5, 2)]
p = ta.P.new(xs)
out = ta.reduce_max(p.roll(1).left().pack())
out.sum().backward()
ta.scatter_sum(torch.zeros(3, 8, device="cuda:0"), torch.tensor([0, 2, 2, 1], device="cuda:0"), torch.randn(4, 8, device="cuda:0"))
torch.cuda.synchronize()
maps = open("/proc/self/maps").read()
assert "librua_hip.so" in maps, "the HIP library is not loaded"
assert "librua_oracle" not in maps, "the oracle library is mapped in a product-only process"
assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), "oracle imported"
print("ok")
''' % root
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stdout + out.stderr


def test_ops_replay_inside_a_hip_graph():
    """The C ABI never allocates, synchronises or reads back (include/rua.h), so once the derived index vectors of
    a container are memoised, its ops can be captured in a HIP graph and replayed on new payload: reduce / roll /
    last / pad / cat over fixed lengths (the launch-bound small-batch case of SURVEY §8d cfg1)."""
    g = torch.Generator().manual_seed(11)
    lens = torch.randint(1, 30, (64,), generator=g)
    n = int(lens.sum())
    static = torch.randn(n, 32, generator=g).to(DEV)
    c = ta.with_host_sizes(static, lens)

    def ops(cc):
        p = cc.pack()
        return ta.reduce_sum(p), ta.reduce_max(p), p.roll(1).data, p.last(), cc.left(-1.0).data, p.cat().data

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                     # warm-up on a side stream: memoises every index vector
        for _ in range(2):
            ops(c)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = ops(c)
    for seed in (1, 2):
        fresh = torch.randn(n, 32, generator=torch.Generator().manual_seed(seed)).to(DEV)
        static.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        want = ops(ta.with_host_sizes(fresh, lens))
        for got, exp in zip(outs, want):
            assert torch.equal(got, exp)


def test_ops_follow_the_current_stream():
    """Everything is enqueued on torch's CURRENT stream (metadata uploads ride their own upload stream and the current
    one waits for them): a whole chain under a non-default stream, consumed on the default stream after the usual
    wait_stream, and many back-to-back steps that recycle the pinned staging ring."""
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(1, 50, (300,), generator=g)
    data = torch.randn(int(lens.sum()), 24, generator=g)
    exp_sum = torch.stack([x.sum(0) for x in torch.split(data, lens.tolist())])
    s1 = torch.cuda.Stream()
    dd = data.to(DEV)
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        c = ta.with_host_sizes(dd, lens)
        p = c.pack()
        out = ta.reduce_sum(p)
        back = p.roll(2).roll(-2).cat().data
    torch.cuda.current_stream().wait_stream(s1)
    torch.testing.assert_close(out.cpu(), exp_sum, rtol=1e-5, atol=1e-5)
    assert torch.equal(back, dd)
    for p_ in (p.data, out, back):
        p_.record_stream(torch.cuda.current_stream())
    # 100 steps with fresh lengths each (3 uploads per step through a 32-slot ring), no sync in between
    outs = []
    for i in range(100):
        li = torch.roll(lens, i)
        ci = ta.with_host_sizes(dd, li)
        outs.append((li, ta.reduce_sum(ci.pack())))
    torch.cuda.synchronize()
    for li, o in outs[::7]:
        ref = torch.stack([x.sum(0) for x in torch.split(data, li.tolist())])
        torch.testing.assert_close(o.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_no_cyclic_garbage_per_step():
    """The memos and plans an op leaves behind must be freed by reference counting alone: cyclic garbage created at
    every step pushes CPython into full collections (tens of ms with torch's heap to scan)."""
    import gc
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(1, 40, (200,), generator=g)
    data = torch.randn(int(lens.sum()), 16, generator=g).to(DEV)
    idx = torch.randint(0, 200, (data.size(0),), generator=g).to(DEV)

    def step():
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        out = ta.reduce_sum(p) + ta.reduce_max(c.left()) + ta.segment_logsumexp(c.data, c.token_sizes)
        q = p.roll(1).rev().cat().right().pack()
        x = data.clone().requires_grad_(True)
        z = ta.C(x, c.token_sizes).pack()
        (ta.reduce_max(z).sum() + z.roll(2).left().data.sum()).backward()
        s = ta.scatter_sum(torch.zeros(200, 16, device=DEV), idx, data)
        xs = c.split()
        return out, q, x.grad, s, ta.C.new(xs).last(), p.head(1), c.trunc((0, 0)), c.bmask()

    for _ in range(3):
        step()
    gc.collect()
    gc.disable()
    try:
        for _ in range(3):
            keep = step()
        del keep
        assert gc.collect() == 0
    finally:
        gc.enable()


def test_chains_with_a_host_mirror_never_sync():
    """Lengths that came from the host (C.new / with_host_sizes) are mirrored, and the mirror travels through
    pack / cat / pad / roll / trunc: no op of such a chain may read anything back from the device (the reference pays a
    blocking .item() or .cpu() in every size() and pack_view, layout/cat.py:61-66, core/view.py:48)."""
    g = torch.Generator().manual_seed(9)
    lens = torch.randint(2, 30, (500,), generator=g)
    data = torch.randn(int(lens.sum()), 8, generator=g).to(DEV)
    c = ta.with_host_sizes(data, lens)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        p = c.pack()
        c2 = p.cat()
        p2 = c2.pack()                      # the mirror came along through the PackedSequence
        l = c2.left(0.5)
        p3 = l.pack()
        r = p3.roll(3).rev().right()
        outs = (ta.reduce_sum(p2), ta.reduce_max(l), ta.reduce_logsumexp(r), p3.last(), c2.head(2).data, c2.trunc((1, 0)).data)
        size = (c2.size(), l.size(), p2.size())
        with pytest.raises(RuntimeError):   # the detector itself works: a device-only length vector must be read back
            ta.C(data, lens.to(DEV, non_blocking=True)).pack()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert torch.equal(c2.data, data) and torch.equal(p2.data, p.data) and size[0][:2] == (500, int(lens.max()))
    assert all(bool(torch.isfinite(o).all()) for o in outs)


def test_ops_run_on_the_stream_and_device_of_their_tensors():
    """VERDICT r4 #8.  Two streams of one card: work enqueued under `torch.cuda.stream(side)` lands on `side` (ordered
    after a delay kernel there) and equals the main stream's result.  Two cards (self-skipping part): tensors on cuda:1
    while cuda:0 is current — torch runs such ops where the tensors live, and so does this library."""
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(1, 40, (300,), generator=g)
    data = torch.randn(int(lens.sum()), 24, generator=g)
    c = ta.C(data.to(DEV), lens.to(DEV))
    want = ta.reduce_max(c.pack())
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        torch.cuda._sleep(20_000_000)                                   # the side stream is busy: its work is ordered behind this
        got = ta.reduce_max(ta.C(c.data, c.token_sizes).pack())
        assert not side.query()                                         # ... and was enqueued there, not on the main stream
    side.synchronize()
    assert torch.equal(got, want)
    if torch.cuda.device_count() >= 2:
        d1 = torch.device('cuda', 1)
        assert torch.cuda.current_device() == 0
        c1 = ta.C(data.to(d1), lens.to(d1))
        out = ta.reduce_max(c1.pack())
        assert out.device == d1 and torch.cuda.current_device() == 0
        assert torch.equal(out.cpu(), want.cpu())
        assert torch.equal(c1.left().data.cpu(), c.left().data.cpu())
