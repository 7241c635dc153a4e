"""What would the reference's OWN path cost on this GPU?  The reference is a composition of stock ATen ops; this
script issues the same kind of composition with plain PyTorch-ROCm ops on the MI355X (host sort, B x T int64 mask,
repeat_interleave pointers, aten::index gathers, torch.segment_reduce) for pack -> cat -> segment_sum at the
north-star shape, and times it next to torchrua_amd.  It is a measurement aid written for this repository from the
call stacks in SURVEY.md §3 — not reference code, and not part of the product or of the tests."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
B, H = int(os.environ.get('B', 65536)), int(os.environ.get('H', 512))
g = torch.Generator().manual_seed(5)
lens_host = torch.randint(8, 513, (B,), generator=g)
N = int(lens_host.sum())
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
lens = lens_host.to(dev)


def stock_pack(data, lens):
    """C -> P with stock ops, following SURVEY.md §3.1 (sizes, ptr, mask, host sort, index gather)."""
    T = int(lens.max().item())                                          # size(): blocking .item()
    _, sorted_idx = torch.sort(lens.detach().cpu(), descending=True)    # host sort
    sorted_idx = sorted_idx.to(dev)
    unsorted_idx = torch.empty_like(sorted_idx)
    unsorted_idx[sorted_idx] = torch.arange(B, device=dev)
    off = torch.cumsum(lens, 0) - lens
    batch_ptr = torch.repeat_interleave(torch.arange(B, device=dev), lens)
    token_ptr = torch.arange(N, device=dev) - torch.repeat_interleave(off, lens)
    mask = torch.zeros((B, T), dtype=torch.long, device=dev)
    mask[batch_ptr, token_ptr] = 1
    batch_sizes = mask.sum(0).cpu()                                     # D2H, as PackedSequence wants it
    bsz = batch_sizes.to(dev)
    boff = torch.cumsum(bsz, 0) - bsz
    p_token = torch.repeat_interleave(torch.arange(T, device=dev), bsz)
    p_rank = torch.arange(N, device=dev) - torch.repeat_interleave(boff, bsz)
    key = off[sorted_idx[p_rank]] + p_token
    return data[key], batch_sizes, sorted_idx, unsorted_idx, boff, (batch_ptr, token_ptr)


def stock_cat_reduce(pdata, lens, unsorted_idx, boff, ptr):
    """P -> C -> per-sequence sum (SURVEY.md §8d spelling A) with stock ops."""
    batch_ptr, token_ptr = ptr
    key = unsorted_idx[batch_ptr] + boff[token_ptr]
    c = pdata[key]
    return torch.segment_reduce(c, 'sum', lengths=lens, unsafe=True)


def timed(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, out


def stock():
    pdata, bs, srt, uns, boff, ptr = stock_pack(data, lens)
    return pdata, stock_cat_reduce(pdata, lens, uns, boff, ptr)


def ours():
    p = ta.with_host_sizes(data, lens_host).pack()
    return p.data, ta.reduce_sum(p)


ms_stock, (p_stock, r_stock) = timed(stock)
ms_ours, (p_ours, r_ours) = timed(ours, iters=10)
assert torch.equal(p_stock, p_ours), 'pack payloads differ'
err = (r_stock.float() - r_ours.float()).abs().max().item()
print(f'shape: B={B} H={H} bf16, N={N} rows ({N * H * 2 / 1e9:.2f} GB)')
print(f'stock PyTorch-ROCm composition : {ms_stock:9.2f} ms per pack->cat->segment_sum')
print(f'torchrua_amd (pack, reduce_sum) : {ms_ours:9.2f} ms   ({ms_stock / ms_ours:.1f}x)')
print(f'max |sum difference| = {err:.4f} (torch.segment_reduce accumulates bf16 inputs its own way; outputs are bf16)')
