"""Developer probe: where the host time of pack() with device-only lengths goes (core._pack_meta_overlapped), stage by
stage, with the GPU idle (perf_counter around each stage; median over the steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchrua_amd as ta  # noqa: E402
from torchrua_amd import _lib as K  # noqa: E402
from torchrua_amd import _meta as M  # noqa: E402

dev = torch.device('cuda:0')
lib = K.load()
B = 65536
lens_h = torch.randint(8, 513, (B,), generator=torch.Generator().manual_seed(5))
threads = M.host_sort_threads()
print('threads', threads)
rows = []
for step in range(40):
    lens = lens_h.to(dev)
    torch.cuda.synchronize()
    time.sleep(0.008)                    # the helper and the workers have gone to sleep, as between two steps
    t = [time.perf_counter()]
    host = M._read_back(lens)
    t.append(time.perf_counter())
    staged_order = torch.empty(B, dtype=torch.long, pin_memory=True)
    t.append(time.perf_counter())
    lib.rua_host_sort_desc_begin(host.data_ptr(), B, staged_order.data_ptr(), threads)
    t.append(time.perf_counter())
    T = int(host.numpy().max())
    batch_sizes = M.batch_sizes_from_host_lens(host, T)
    t.append(time.perf_counter())
    staged = torch.empty(2 * T + B, dtype=torch.long, pin_memory=True)
    staged[:T].copy_(batch_sizes)
    lib.rua_host_pack_scans(host.data_ptr(), B, batch_sizes.data_ptr(), T, staged.data_ptr() + 8 * T, staged.data_ptr() + 16 * T)
    t.append(time.perf_counter())
    unsorted = torch.empty(B, dtype=torch.long, device=dev)
    sorted_indices = torch.empty(B, dtype=torch.long, device=dev)
    buf = torch.empty(2 * T + B, dtype=torch.long, device=dev)
    t.append(time.perf_counter())
    lib.rua_host_sort_desc_end()
    t.append(time.perf_counter())
    sorted_indices.copy_(staged_order, non_blocking=True)
    buf.copy_(staged, non_blocking=True)
    t.append(time.perf_counter())
    lib.rua_pack_meta(None, K.ptr(sorted_indices), B, 0, K.ptr(unsorted), None, K.stream_ptr(dev))
    t.append(time.perf_counter())
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    # the sort alone, synchronous, threads asleep
    time.sleep(0.008)
    s0 = time.perf_counter()
    lib.rua_host_sort_desc(host.data_ptr(), B, staged_order.data_ptr(), threads)
    s1 = time.perf_counter()
    rows.append([(b - a) * 1e6 for a, b in zip(t, t[1:])] + [(s1 - s0) * 1e6])
names = ['read_back', 'pinned alloc', 'begin()', 'max + batch_sizes', 'staging + scans', '3 device allocs', 'end() wait', '2 async copies', 'pack_meta launch', 'final sync', 'sync sort (cold threads)']
med = [sorted(c)[len(c) // 2] for c in zip(*rows[4:])]
print('  '.join(f'{n} {v:.0f}' for n, v in zip(names, med)), f'| sum to launch {sum(med[:9]):.0f} us')
