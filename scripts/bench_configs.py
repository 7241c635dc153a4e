"""Developer bench (not the driver's bench.py): every BASELINE.json config + a few conversions,
algorithmic bytes per SURVEY.md §8(d), HIP-event timing, one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')


def inputs(seed, B, lo, hi, H, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    n = int(lens.sum())
    data = torch.empty((n, H), dtype=dtype, device=dev)
    step = 1 << 21
    for a in range(0, n, step):
        data[a:min(n, a + step)] = torch.randn((min(n, a + step) - a, H), device=dev)
    return lens, data


def timeit(name, fn, nbytes, iters=10):
    """Median of `iters` HIP-event timings.  Calls shorter than a millisecond are timed in bursts of 8 back-to-back
    launches (one event pair around the burst): with a drained queue in front of every call, a 100 us kernel would be
    charged ~8 us of launch latency that a running pipeline never sees; the single-call latency is printed beside it."""
    fn()
    torch.cuda.synchronize()

    def once(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    single = sorted(once(1) for _ in range(iters))[iters // 2]
    ms, note = single, ''
    if single < 1.0:
        ms = sorted(once(8) for _ in range(iters))[iters // 2]
        note = f'   [single call from an idle queue: {single * 1e3:.0f} us]'
    print(f'{name:44s} {ms:9.3f} ms  {nbytes / 1e9:8.2f} GB  {nbytes / ms / 1e9:6.2f} TB/s  ({nbytes / ms / 1e9 / 8 * 100:4.1f} % of 8 TB/s){note}')


def main():
    e = 2
    print('--- cfg2: B=4096 U(8,512) H=256 bf16')
    lens, data = inputs(2, 4096, 8, 512, 256)
    c = ta.with_host_sizes(data, lens)
    N, H, B = data.size(0), 256, 4096
    p = c.pack()
    timeit('c.pack()', lambda: c.pack(), 2 * N * H * e)
    timeit('reduce_sum(p)', lambda: ta.reduce_sum(p), N * H * e + B * H * e)
    timeit('p.cat()', lambda: p.cat(), 2 * N * H * e)
    timeit('segment_sum(c)', lambda: ta.segment_sum(c.data, c.token_sizes), N * H * e + B * H * e)
    T = int(lens.max())
    timeit('c.left()  (pad)', lambda: c.left(), N * H * e + B * T * H * e)

    print('--- cfg3: 16384 segments U(1,64) H=512 bf16')
    lens, data = inputs(3, 16384, 1, 64, 512)
    ld = lens.to(dev)
    N, H, S = data.size(0), 512, 16384
    for name in ('max', 'sum', 'logsumexp'):
        fn = getattr(ta, f'segment_{name}')
        timeit(f'segment_{name}', lambda: fn(data, ld), N * H * e + S * H * e)

    print('--- north star: B=65536 U(8,512) H=512 bf16')
    lens, data = inputs(5, 65536, 8, 512, 512)
    c = ta.with_host_sizes(data, lens)
    N, H, B = data.size(0), 512, 65536
    p = c.pack()
    timeit('c.pack()', lambda: c.pack(), 2 * N * H * e)
    timeit('reduce_sum(p)', lambda: ta.reduce_sum(p), N * H * e + B * H * e)
    timeit('reduce_max(p)', lambda: ta.reduce_max(p), N * H * e + B * H * e)
    timeit('reduce_logsumexp(p)', lambda: ta.reduce_logsumexp(p), N * H * e + B * H * e)
    timeit('p.cat()', lambda: p.cat(), 2 * N * H * e)
    timeit('c.roll(0) (streaming copy through the mover)', lambda: c.roll(0), 2 * N * H * e)
    timeit('p.roll(0)', lambda: p.roll(0), 2 * N * H * e)
    outbuf = torch.empty_like(data)
    timeit('torch copy_', lambda: outbuf.copy_(data), 2 * N * H * e)
    del outbuf
    timeit('pack_reduce(c) fused', lambda: ta.pack_reduce(c, 'sum'), 2 * N * H * e + B * H * e)
    timeit('segment_sum(c)', lambda: ta.segment_sum(c.data, c.token_sizes), N * H * e + B * H * e)
    T = int(lens.max())
    timeit('c.left()  (pad)', lambda: c.left(), N * H * e + B * T * H * e)
    timeit('p.left()  (pad from pack)', lambda: p.left(), N * H * e + B * T * H * e)
    lft = c.left()
    timeit('l.cat()   (unpad)', lambda: lft.cat(), 2 * N * H * e)
    timeit('l.pack()', lambda: lft.pack(), 2 * N * H * e)
    # ---- the integer kernels at the same shape (bytes = what they write; all int64 like the reference's)
    from torchrua_amd import _ops as O
    timeit('c.ptr()  (batch_ptr, token_ptr)', lambda: c.ptr(), 2 * N * 8)
    timeit('p.ptr()', lambda: p.ptr(), 2 * N * 8)
    timeit('c.idx()', lambda: c.idx(), N * 8)
    timeit('l.idx()  (flat index of a padded batch)', lambda: lft.idx(), N * 8)
    timeit('get_mask(c)  [B, T] int64', lambda: ta.get_mask(c), B * T * 8)
    timeit('c.bmask()    [B, T] bool', lambda: c.bmask(), B * T)
    bp = c.ptr()[0]
    shuffled = bp[torch.randperm(N, device=dev)]
    timeit('index_buckets (17 M keys, 65 536 destinations)', lambda: O.index_buckets(shuffled, B), 2 * N * 8)
    del bp, shuffled
    del lft, p, c, data

    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    B = 65536 if free > 215e9 else 16384
    print(f'--- cfg4: B={B} U(16,1024) H=1024 bf16   ({free / 1e9:.0f} GB free before allocation'
          f'{"" if B == 65536 else "; REDUCED from the BASELINE size 65536"})')
    lens, data = inputs(4, B, 16, 1024, 1024)
    N, H = data.size(0), 1024
    p = ta.with_host_sizes(data, lens).pack()
    del data
    timeit('p.roll(1)', lambda: p.roll(1), 2 * N * H * e, iters=5)
    timeit('p.last()', lambda: p.last(), 2 * B * H * e)
    timeit('p.head(16) (view)', lambda: p.head(16), 1)
    timeit('p.rev()', lambda: p.rev(), 2 * N * H * e, iters=5)
    T4 = int(lens.max())
    timeit('p.ptr()      (cfg4)', lambda: p.ptr(), 2 * N * 8, iters=5)
    timeit('p.idx()      (cfg4)', lambda: p.idx(), N * 8, iters=5)
    timeit('get_mask(p)  (cfg4) [B, T] int64', lambda: ta.get_mask(p), B * T4 * 8, iters=5)


if __name__ == '__main__':
    main()
