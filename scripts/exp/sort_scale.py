# rua_host_sort_desc of 65 536 lengths by thread count (profiles/r04_host_sort_scaling.txt; its spawn_min column came from a
# build with a temporary RUA_HOST_SORT_SPAWN_MIN knob: 4 096 stayed)
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrua_amd import _lib as L
lib = L.load()
g = torch.Generator().manual_seed(1)
keys = torch.randint(8, 513, (65536,), generator=g)
out = torch.empty_like(keys)
ref = torch.sort(keys, descending=True)[1]
for th in (4, 6, 8, 12, 16):
    for _ in range(20): lib.rua_host_sort_desc(keys.data_ptr(), keys.numel(), out.data_ptr(), th)
    assert torch.equal(out, ref)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); lib.rua_host_sort_desc(keys.data_ptr(), keys.numel(), out.data_ptr(), th); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f'spawn_min={os.environ.get("RUA_HOST_SORT_SPAWN_MIN","4096"):>5} threads={th:2d}  median {ts[100]*1e6:7.1f} us  p10 {ts[20]*1e6:7.1f}', flush=True)
