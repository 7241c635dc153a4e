import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torchrua_amd as ta
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
lens = torch.randint(8, 513, (65536,), generator=g)
N = int(lens.sum())
data = torch.randn(N, 512, device=dev, dtype=torch.bfloat16)
p = ta.with_host_sizes(data, lens).pack()
idx = p.ptr()[0]
zeros = torch.zeros(65536, 512, device=dev, dtype=torch.bfloat16)
def t(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:40s} {e0.elapsed_time(e1)/n:8.3f} ms')
    return out
a = t('scatter_sum(zeros, p.ptr()[0], p.data)  [spelling B]', lambda: ta.scatter_sum(zeros, idx, p.data))
b = t('reduce_sum(p)', lambda: ta.reduce_sum(p))
from torchrua_amd.reduce import _buckets
t('  of which bucketing (radix sort)', lambda: _buckets(idx, 65536))
print('equal to reduce_sum(p):', torch.equal(a, b))
t('torch.index_add (stock)', lambda: torch.index_add(zeros, 0, idx, p.data))
