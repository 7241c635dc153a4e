import time, torch
dev = torch.device('cuda:0')
side = torch.cuda.Stream(dev)
cur = torch.cuda.current_stream(dev)
def t(name, fn, n=20000):
    for _ in range(1000): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f'{name:40s} {(time.perf_counter() - t0) / n * 1e6:6.2f} us')
def ctx():
    with torch.cuda.stream(side): pass
def ss():
    torch.cuda.set_stream(side); torch.cuda.set_stream(cur)
def raw():
    torch._C._cuda_setStream(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
    torch._C._cuda_setStream(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
t('with torch.cuda.stream(side)', ctx)
t('set_stream x2', ss)
t('_cuda_setStream x2', raw)
t('current_stream(dev)', lambda: torch.cuda.current_stream(dev))
t('_cuda_getCurrentRawStream', lambda: torch._C._cuda_getCurrentRawStream(0))
t('Event()+record', lambda: torch.cuda.Event().record(cur))
e = torch.cuda.Event(); e.record(cur)
t('cur.wait_event', lambda: cur.wait_event(e))
x = torch.empty(10, device=dev)
t('record_stream', lambda: x.record_stream(cur))
t('torch.empty(dev)', lambda: torch.empty(4096, dtype=torch.long, device=dev))
