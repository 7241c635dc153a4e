"""Developer probe (not a test): raw C-ABI calls on the GPU box, checked against stock torch ops,
plus first bandwidth numbers at the north-star shape.  Run: gpurun -- python scripts/gpu_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrua_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
print('device', torch.cuda.get_device_name(0), 'target', lib.rua_build_target())
S = lambda: L.stream_ptr(dev)


def scan(x):
    n = x.numel()
    out = torch.empty_like(x)
    tot = torch.empty(1, dtype=torch.long, device=dev)
    ws = torch.empty(lib.rua_scan_ws_elems(n), dtype=torch.long, device=dev)
    L.check(lib.rua_exclusive_scan_i64(x.data_ptr(), out.data_ptr(), tot.data_ptr(), n, ws.data_ptr(), S()), 'scan')
    return out, tot


for n in (1, 5, 2048, 2049, 65536, 1000003):
    x = torch.randint(0, 1000, (n,), device=dev)
    o, t = scan(x)
    ref = x.cumsum(0) - x
    assert torch.equal(o, ref), n
    assert t.item() == x.sum().item()
print('scan ok')


def build(B, lo, hi, H, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    N = int(lens.sum())
    data = torch.randn(N, H, device=dev, dtype=torch.float32).to(dtype)
    return lens, data


def pack_reduce(lens_cpu, data, check=True, iters=0):
    B = lens_cpu.numel()
    N, H = data.shape
    T = int(lens_cpu.max())
    lens = lens_cpu.to(dev)
    _, sorted_cpu = torch.sort(lens_cpu, descending=True)
    sorted_idx = sorted_cpu.to(dev)
    off, _ = scan(lens)
    unsorted = torch.empty(B, dtype=torch.long, device=dev)
    bsz = torch.empty(T, dtype=torch.long, device=dev)
    L.check(lib.rua_pack_meta(lens.data_ptr(), sorted_idx.data_ptr(), B, T, unsorted.data_ptr(), bsz.data_ptr(), S()), 'meta')
    boff, _ = scan(bsz)
    src = L.RuaLayout(kind=L.CAT, n_rows=N, B=B, lens=lens.data_ptr(), off=off.data_ptr())
    dst = L.RuaLayout(kind=L.PACK, n_rows=N, B=B, lens=lens.data_ptr(), boff=boff.data_ptr(), T=T,
                      sorted=sorted_idx.data_ptr(), unsorted=unsorted.data_ptr())
    pdata = torch.empty_like(data)
    rb = H * data.element_size()
    out = torch.empty(B, H, dtype=data.dtype, device=dev)

    def run_move():
        L.check(lib.rua_move_rows(ctypes.byref(dst), ctypes.byref(src), L.T_SHIFT, 0, pdata.data_ptr(), data.data_ptr(),
                                  rb, None, 0, S()), 'move')

    def run_reduce():
        L.check(lib.rua_segment_reduce(ctypes.byref(dst), None, pdata.data_ptr(), out.data_ptr(), H,
                                       L.DTYPES[data.dtype], L.SUM, 0, 0, None, S()), 'reduce')

    run_move()
    run_reduce()
    torch.cuda.synchronize()
    if check:
        from torch.nn.utils.rnn import pack_sequence
        seqs = list(torch.split(data, lens_cpu.tolist()))
        p = pack_sequence(seqs, enforce_sorted=False)
        assert torch.equal(p.batch_sizes, bsz.cpu())
        assert torch.equal(p.sorted_indices, sorted_idx)
        assert torch.equal(p.unsorted_indices, unsorted)
        assert torch.equal(p.data, pdata), 'pack payload'
        ref = torch.stack([s.float().sum(0) for s in seqs])
        err = (out.float() - ref).abs().max().item()
        print('  reduce max abs err', err, 'ref scale', ref.abs().max().item())
    if iters:
        for name, fn, nbytes in (('move', run_move, 2 * N * rb), ('reduce', run_reduce, N * rb + B * rb)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f'  {name}: {ms:.3f} ms  {nbytes / ms / 1e9:.3f} TB/s ({nbytes / 1e9:.2f} GB)')


print('small f32 H=7')
pack_reduce(*build(37, 1, 9, 7, torch.float32))
print('small bf16 H=64')
pack_reduce(*build(100, 1, 50, 64, torch.bfloat16))
print('cfg2 bf16 H=256')
pack_reduce(*build(4096, 8, 512, 256, torch.bfloat16, seed=2), iters=10)
print('NS bf16 H=512 B=65536')
t = time.time()
lens, data = build(65536, 8, 512, 512, torch.bfloat16, seed=5)
print('  built in', time.time() - t, 's; N =', data.shape[0])
pack_reduce(lens, data, check=False, iters=10)
print('PROBE OK')
