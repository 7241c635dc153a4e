"""Developer probe (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): one launch of each narrow-row kernel at
32-byte rows (H = 16 bf16, ~8 GB payload) so that the counters can be set against the algorithmic bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchrua_amd as ta  # noqa: E402

dev = torch.device('cuda:0')
H = int(os.environ.get('RUA_PROBE_H', 16))
rows = int(8e9 / (H * 2))
B = max(1024, rows // 260)
g = torch.Generator().manual_seed(H)
lens = torch.randint(8, 513, (B,), generator=g)
N = int(lens.sum())
data = torch.randn(N, H, device=dev, dtype=torch.bfloat16)
c = ta.with_host_sizes(data, lens)
p = c.pack()
q = p.cat()
r = p.roll(1)
s = ta.reduce_sum(p)
t = ta.segment_sum(c.data, c.token_sizes)
torch.cuda.synchronize()
print('payload bytes', N * H * 2)
