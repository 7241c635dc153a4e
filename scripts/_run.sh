set -e
python -m pytest tests/test_host_sort.py -q 2>&1 | tail -2
python - <<'PY'
import torch, time, numpy as np
from torchrua_amd import _lib as L, _meta as M
lib=L.load()
l=torch.randint(8,513,(65536,),generator=torch.Generator().manual_seed(5))
out=torch.empty_like(l)
torch.set_num_threads(1)
for th in (1,2,4,8,16):
    lib.rua_host_sort_desc(l.data_ptr(),65536,out.data_ptr(),th)
    t=time.perf_counter()
    for _ in range(50): lib.rua_host_sort_desc(l.data_ptr(),65536,out.data_ptr(),th)
    print(th,'threads ms',(time.perf_counter()-t)/50*1e3)
t=time.perf_counter()
for _ in range(20): torch.sort(l,descending=True)
print('torch ms',(time.perf_counter()-t)/20*1e3)
t=time.perf_counter(); M.host_sort_desc(l); print('decide+first ms',(time.perf_counter()-t)*1e3,'threads',M._host_sort_threads)
PY
python scripts/stall_probe.py copy 2>&1 | tail -8
python bench.py --steps 20 --warmup 3 --trace-host > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_a.json'))
print(d['value'], d['ms_per_step'], d['device_lens'], d['fused_pack_reduce']['ms_per_step'], d['roofline']['avg_ms'], d['reduce_kernel']['avg_ms'])
PY
tail -2 gpurun_out/bench_a.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
