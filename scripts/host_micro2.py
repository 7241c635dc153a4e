"""Developer probe: what the small host-side building blocks of a launch cost on the GPU box."""
import time
import torch

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
x = torch.empty(8, device=dev)


def t(name, fn, n=20000):
    for _ in range(200):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f'{name:46s} {(time.perf_counter() - t0) / n * 1e6:7.2f} us')


cur = torch.cuda.current_stream(dev)
ev = torch.cuda.Event()
side = torch.cuda.Stream(dev)
t('torch.cuda.Event()', lambda: torch.cuda.Event())
t('Event() + record(cur)', lambda: torch.cuda.Event().record(cur), 5000)
t('reused event .record(cur)', lambda: ev.record(cur), 5000)
t('cur.wait_event(ev)', lambda: cur.wait_event(ev), 5000)
t('torch.cuda.current_stream(dev)', lambda: torch.cuda.current_stream(dev))
t('torch._C._cuda_getCurrentRawStream(0)', lambda: torch._C._cuda_getCurrentRawStream(0))
t('torch.empty(4096, long, device)', lambda: torch.empty(4096, dtype=torch.long, device=dev))
t('torch.cuda.set_stream(side); set_stream(cur)', lambda: (torch.cuda.set_stream(side), torch.cuda.set_stream(cur)))
t('torch.cuda.is_current_stream_capturing()', lambda: torch.cuda.is_current_stream_capturing())
from torch.nn.utils.rnn import PackedSequence
bs = torch.ones(4, dtype=torch.long)
t('PackedSequence(...)', lambda: PackedSequence(x, bs, None, None))
torch.cuda.synchronize()
