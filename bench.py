"""bench.py — the north-star measurement (BASELINE.json): pack -> reduce over 65 536 variable-length
sequences, hidden 512, bf16, per GPU; M elements/s whole-job + % of the HBM roofline.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch already resident in HBM:
    c = C(data, token_sizes)            # token_sizes handed over from the host, as C.new does
    p = c.pack()                        # host sort (reference's own call) + K1/K3 metadata + row mover
    out = reduce_sum(p)                 # segmented reduce over the PackedSequence -> [B, H]
    (N > 1: one RCCL all-gather of `out`; sequences are sharded, payload never crosses xGMI)
Nothing is cached between steps: every step uploads the lengths again, re-sorts, re-scans, re-moves.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the pack row mover:
algorithmic bytes 2*N*H*e + 8*(3B+T)), timed with HIP events on the launch stream
inside the timed region.  `cpu_baseline` times the oracle port on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=65536, help='sequences per GPU')
    ap.add_argument('--hidden', type=int, default=512)
    ap.add_argument('--lo', type=int, default=8)
    ap.add_argument('--hi', type=int, default=512)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--trace-host', action='store_true', help='print per-step host enqueue time to stderr')
    ap.add_argument('--cpu-sample', type=int, default=16384, help='sequences in the CPU-baseline sample (16 384 = a quarter of the batch: ~10 s of host work for the two CPU legs)')
    return ap.parse_args()


def make_inputs(args, rank, dev):
    """SURVEY.md §8(d) recipe; the payload is drawn on the device (17.45 GB per rank at the default shape)."""
    g = torch.Generator().manual_seed(5 + rank)
    lens = torch.randint(args.lo, args.hi + 1, (args.batch,), generator=g)
    n = int(lens.sum())
    dg = torch.Generator(device=dev).manual_seed(1005 + rank)
    data = torch.empty((n, args.hidden), dtype=torch.bfloat16, device=dev)
    chunk = 1 << 21
    for lo in range(0, n, chunk):   # bounded fp32 temporaries
        hi = min(n, lo + chunk)
        data[lo:hi] = torch.randn((hi - lo, args.hidden), generator=dg, device=dev, dtype=torch.float32)
    return lens, data


class KernelTimer:
    """HIP events around named kernel launches, recorded on the stream the kernels are launched on."""

    def __init__(self, names):
        self.names, self.pairs, self.open = set(names), {}, {}
        self.enabled = False

    def __call__(self, name, begin):
        if not self.enabled or name not in self.names:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if begin:
            self.open[name] = ev
        else:
            self.pairs.setdefault(name, []).append((self.open.pop(name), ev))

    def mean_ms(self, name):
        p = self.pairs.get(name, [])
        return sum(a.elapsed_time(b) for a, b in p) / len(p) if p else None


def cpu_baseline(args):
    """The oracle port (oracle/rua_oracle.{py,c}: numpy + C/OpenMP restatement of the reference's
    algorithm) on this box's host cores, over a bounded sample of the same workload."""
    import numpy as np

    from oracle import rua_oracle as orc
    cores = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    B = args.cpu_sample
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(args.lo, args.hi + 1, (B,), generator=g)
    n = int(lens.sum())
    data = torch.randn((n, args.hidden), generator=g).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    lens_np = lens.numpy()
    import ctypes
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        srt = torch.sort(lens, descending=True)[1].numpy()                  # core/view.py:48
        p = orc.to_pack(orc.C(data, lens_np), srt)                          # core/cast.py:41-49
        c = orc.to_cat(p)                                                   # core/cast.py:8-10  (spelling A,
        out = np.empty((B, args.hidden), np.float32)                        #  SURVEY.md §8d) + reduce.py:44
        off = orc.get_offsets(c.token_sizes)
        orc.lib().orc_segment_sum_bf16(orc._p(c.data), orc._p(c.token_sizes), orc._p(off), ctypes.c_int64(B),
                                       ctypes.c_int64(args.hidden), orc._p(out))
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {'value': round(n * args.hidden / t / 1e6, 1), 'unit': 'M elements/s', 'cores': cores, 'kind': 'port',
            'sample': f'{B} sequences len~U({args.lo},{args.hi}) hidden={args.hidden} bf16 ({n} rows), '
                      f'pack -> cat -> segment_sum, median of 5, {t:.2f} s each'}


def parity_leg(ta, data, lens_host, out_bf16, n_sample=256):
    """Untimed: what the kernels' floating point actually achieves at this shape, against the oracle (checker only).
    A strided sample of `n_sample` sequences of the batch is copied out; the oracle folds it the reference's way
    (torch.segment_reduce's sequential fp32 fold over x.float(), reduce.py:44-61) and exactly (fp64).  Reported:
      * the graded bf16 output of the pipeline against the oracle's fp32 sums rounded to bf16, in bf16 ulps;
      * the kernels on the same rows as fp32 payload against the oracle: |got - ref| / sum|x| per element for the sums
        (the scale rounding errors of a sum live on) and plain relative error for logsumexp; max must be exact;
      * the reference's OWN sequential fp32 fold against fp64 in the same measure, for scale.
    north_star's bar is 1e-5 relative."""
    import numpy as np

    from oracle import rua_oracle as orc
    B = lens_host.numel()
    pick = torch.arange(0, B, max(1, B // n_sample))[:n_sample]
    off = torch.cumsum(lens_host, 0) - lens_host
    rows = torch.cat([torch.arange(int(off[b]), int(off[b] + lens_host[b])) for b in pick.tolist()])
    lens_s = lens_host[pick].contiguous()
    x_dev = data[rows.to(data.device)]                              # [n_s, H] bf16 on the device
    x32 = x_dev.float()
    x_np = x32.cpu().numpy()
    lens_np = lens_s.numpy()
    ref32 = {'sum': orc.segment_sum(x_np, lens_np), 'max': orc.segment_max(x_np, lens_np),
             'logsumexp': orc.segment_logsumexp(x_np, lens_np)}
    x64 = x_np.astype(np.float64)
    ref64_sum = orc.segment_sum(x64, lens_np)
    ref64_lse = orc.segment_logsumexp(x64, lens_np)
    sum_abs = orc.segment_sum(np.abs(x64), lens_np)
    lens_dev = lens_s.to(data.device)
    got = {'sum': ta.segment_sum(x32, lens_dev).cpu().numpy(), 'max': ta.segment_max(x32, lens_dev).cpu().numpy(),
           'logsumexp': ta.segment_logsumexp(x32, lens_dev).cpu().numpy()}
    # the graded output: rows of the pipeline's [B, H] bf16 sums for the sampled sequences
    mine_bf16 = out_bf16[pick.to(out_bf16.device)].view(torch.int16).cpu().numpy().astype(np.int32)
    want_bf16 = torch.from_numpy(ref64_sum).to(torch.bfloat16).view(torch.int16).numpy().astype(np.int32)

    def ordered(bits):          # sign-magnitude -> monotone integers, so that a difference counts ulps
        return np.where(bits < 0, -(bits & 0x7fff), bits)
    ulps = np.abs(ordered(mine_bf16) - ordered(want_bf16))
    return {
        'sample': f'{pick.numel()} sequences (every {max(1, B // n_sample)}th) of the batch, {rows.numel()} rows',
        'pipeline_bf16_sum_vs_exact_rounded_to_bf16_max_ulps': int(ulps.max()),
        'pipeline_bf16_sum_frac_bit_identical': round(float((ulps == 0).mean()), 4),
        'sum_f32_vs_reference_fold_over_sum_abs': float(np.max(np.abs(got['sum'] - ref32['sum']) / sum_abs)),
        'sum_f32_vs_fp64_over_sum_abs': float(np.max(np.abs(got['sum'] - ref64_sum) / sum_abs)),
        'reference_fold_f32_vs_fp64_over_sum_abs': float(np.max(np.abs(ref32['sum'] - ref64_sum) / sum_abs)),
        'sum_f32_vs_fp64_max_rel': float(np.max(np.abs(got['sum'] - ref64_sum) / np.maximum(np.abs(ref64_sum), 1e-30))),
        'reference_fold_f32_vs_fp64_max_rel': float(np.max(np.abs(ref32['sum'] - ref64_sum) / np.maximum(np.abs(ref64_sum), 1e-30))),
        'max_exact': bool(np.array_equal(got['max'], ref32['max'])),
        'logsumexp_f32_vs_reference_max_rel': float(np.max(np.abs(got['logsumexp'] - ref32['logsumexp']) / np.maximum(np.abs(ref32['logsumexp']), 1e-30))),
        'logsumexp_f32_vs_fp64_max_rel': float(np.max(np.abs(got['logsumexp'] - ref64_lse) / np.maximum(np.abs(ref64_lse), 1e-30))),
        'bar': '1e-5 relative (BASELINE.json north_star); integer outputs bit-exact (checked by tests, not here)'}


def _ranges(cpus):
    """[0, 1, 2, 3, 8, 9] -> '0-3,8-9' (the CPU set a rank bound itself to)."""
    out, run = [], []
    for c in sorted(cpus):
        if run and c == run[-1] + 1:
            run.append(c)
        else:
            if run:
                out.append(run)
            run = [c]
    if run:
        out.append(run)
    return ','.join(f'{r[0]}-{r[-1]}' if len(r) > 1 else str(r[0]) for r in out)


def _gpu_numa_node(dev):
    """What parallel.bind_rank_to_cpus reads from sysfs for this card (None when the platform does not say)."""
    try:
        from torchrua_amd import parallel
        return parallel._numa_node_of_gpu(parallel._pci_name(torch.cuda.get_device_properties(dev.index or 0)))
    except Exception:
        return None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one fresh child process per GPU, BEFORE
    anything in this process touches the GPU (the parent never does), and leave with their exit code.  The children
    are this same script with the rendezvous environment torch.distributed.run would have set."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()          # counts devices without initialising the runtime
    if 'RUA_BENCH_DEVICE' not in os.environ and have < n:
        raise SystemExit(f'bench.py: --gpus {n} but this node shows {have} GPU(s)')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    # the host-sort thread count is decided ONCE, here, for all ranks (each would otherwise time 1/2/4/8 threads while
    # its seven neighbours do the same): a rank's share of the cores, at most 4 (the sort stops scaling there)
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    sort_threads = os.environ.get('RUA_HOST_SORT', str(max(1, min(4, cpus // (2 * n)))))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RUA_HOST_SORT=sort_threads)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, pr in list(pending.items()):
            code = pr.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code
                print(f'bench.py: rank {r} exited with {code}; stopping the other ranks', file=sys.stderr)
                for other in pending.values():     # exactly the children started above
                    other.terminate()
        time.sleep(0.05)
    raise SystemExit(rc)


def cpu_baseline_torch(args):
    """The library-independent calls the reference's own tests use as oracles, on this box's host cores
    (SURVEY.md §8d (2)(i)): torch.nn.utils.rnn.pack_sequence(enforce_sorted=False) — tests/expected.py:21-22 —
    then torch.segment_reduce(x.float(), 'sum', lengths) — reduce.py:44-45; fp32 because ATen's CPU kernel
    accumulates bf16 sequentially in bf16.  Same bounded sample as the port."""
    from torch.nn.utils.rnn import pack_sequence
    B = args.cpu_sample
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(args.lo, args.hi + 1, (B,), generator=g)
    n = int(lens.sum())
    data = torch.randn((n, args.hidden), generator=g).to(torch.bfloat16)
    xs = list(torch.split(data, lens.tolist()))              # views: the list form pack_sequence takes
    threads = torch.get_num_threads()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        p = pack_sequence(xs, enforce_sorted=False)
        out = torch.segment_reduce(data.float(), 'sum', lengths=lens, unsafe=True)
        times.append(time.perf_counter() - t0)
    assert p.data.shape == data.shape and out.shape == (B, args.hidden)
    t = sorted(times)[len(times) // 2]
    return {'value': round(n * args.hidden / t / 1e6, 1), 'unit': 'M elements/s', 'cores': threads, 'kind': 'stock torch',
            'sample': f'{B} sequences len~U({args.lo},{args.hi}) hidden={args.hidden} bf16 ({n} rows), '
                      f'pack_sequence(enforce_sorted=False) + segment_reduce(x.float(), sum), torch intra-op threads = '
                      f'{threads}, median of 3, {t:.2f} s each'}


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a '
                         f'different number of ranks than was asked for')
    # rehearsal knobs (1-GPU box): RUA_BENCH_DEVICE pins every rank to one card, RUA_BENCH_BACKEND=gloo replaces RCCL
    dev = torch.device('cuda', int(os.environ.get('RUA_BENCH_DEVICE', local_rank)))
    torch.cuda.set_device(dev)
    # one rank per GPU on one host: every rank keeps to its share of the cores (those next to its card) and caps torch's
    # intra-op pool to it — eight unpinned ranks otherwise run eight host sorts and eight 128-thread pools on every core
    from torchrua_amd.parallel import bind_rank_to_cpus
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    try:
        my_cpus = bind_rank_to_cpus(local_rank, local_world, dev.index)
    except Exception as exc:                     # a host the planner does not understand must never cost the run
        print(f'bench.py: rank {rank}: CPU binding skipped ({exc!r})', file=sys.stderr)
        my_cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    import torch.distributed as dist
    use_dist = world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ)   # under torchrun
    if use_dist:
        backend = os.environ.get('RUA_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import torchrua_amd as ta
    from torchrua_amd import _ops
    from torchrua_amd.parallel import all_gather_rows

    lens_host, data = make_inputs(args, rank, dev)
    B, H = args.batch, args.hidden
    N, T = int(data.size(0)), int(lens_host.max())
    e = data.element_size()

    timer = KernelTimer(['to_pack', 'reduce', 'pack_reduce'])
    _ops.set_kernel_hook(timer)

    def step(host_mirror=True):
        if host_mirror:
            c = ta.with_host_sizes(data, lens_host)          # lengths arrive from the host (as in C.new)
        elif host_mirror is None:
            c = ta.C(data, lens_master.clone())               # lengths a previous kernel left on the device
        else:
            c = ta.C(data, lens_host.to(dev))                 # device-only lengths: pack() must read them back
        p = c.pack()
        out = ta.reduce_sum(p)
        enqueue_s[0] = time.perf_counter()        # the rank's own host work ends here; what follows waits for peers
        if use_dist:
            # ONE RCCL all-gather of [B, H] per step, on RCCL's stream: it overlaps the next step's
            # kernels; at most one is in flight, and the last one is waited for inside the timed region
            if pending:
                pending.pop().wait()
            out, work = all_gather_rows(out, async_op=True)
            pending.append(work)
        return p, out

    pending = []
    enqueue_s = [0.0]
    lens_master = lens_host.to(dev)

    def drain():
        while pending:
            pending.pop().wait()

    def sync():
        drain()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    p = out = None
    for _ in range(max(args.warmup, 2)):   # at least 2: the loop keeps the previous PackedSequence alive, so the
                                           # allocator needs two steps to own both 17 GB buffers
        p, out = step()      # keep the previous result alive exactly like the timed loop does, so the caching
    sync()                   # allocator reaches its steady state (two 17 GB P buffers) before the clock starts
    timer.enabled = True
    t0 = time.perf_counter()
    host_ms = []
    for _ in range(args.steps):
        h0 = time.perf_counter()
        p, out = step()
        host_ms.append((enqueue_s[0] - h0) * 1e3)       # pack() + reduce_sum() enqueued (host sort included)
    sync()
    dt = time.perf_counter() - t0
    if args.trace_host and rank == 0:
        print('host enqueue ms per step:', ' '.join(f'{x:.2f}' for x in host_ms), file=sys.stderr)
        for name in ('to_pack', 'reduce'):
            print(f'{name} kernel ms per step:', ' '.join(f'{a.elapsed_time(b):.2f}' for a, b in timer.pairs.get(name, [])),
                  file=sys.stderr)
    timer.enabled = False

    out_graded = out        # the timed pipeline's [B, H] sums (parity_leg looks at a sample of them)
    # sanity inside the bench: the last step's output is a real PackedSequence and a [B, H] sum
    assert p.data.shape == data.shape and p.batch_sizes.numel() == T and out.shape[-1] == H
    assert out.shape[0] == B * world

    extra = {}
    if world == 1:                     # the same pipeline with device-only lengths (blocking D2H per pack)
        sync()
        k = max(5, args.steps // 2)
        step(host_mirror=False)
        sync()
        t1 = time.perf_counter()
        for _ in range(k):
            p, out = step(host_mirror=False)
        sync()
        dl_ms = (time.perf_counter() - t1) / k * 1e3
        extra['value_device_lens'] = round(N * H / (dl_ms * 1e-3) / 1e6, 1)
        extra['value_reference_signature'] = extra['value_device_lens']     # (the call a user of the reference makes)
        extra['device_lens'] = {
            'call': 'C(data, token_sizes_host.to(device)).pack() -> reduce_sum: the reference\'s own constructor signature; '
                    'every step uploads the lengths from pageable memory, every pack() reads them back (blocking D2H) '
                    'before the host sort',
            'value': extra['value_device_lens'], 'unit': 'M elements/s', 'ms_per_step': round(dl_ms, 4), 'steps': k,
            'frac_of_hbm_peak_wall': round((3.0 * N * H * e + 1.0 * B * H * e + 8.0 * (4 * B + T)) / (dl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # the same with lengths that are PRODUCED on the device (a device-to-device copy stands in for the kernel that
        # computed them: no pageable upload, no implicit synchronisation of the host before pack() asks for them)
        step(host_mirror=None)
        sync()
        t1 = time.perf_counter()
        for _ in range(k):
            p, out = step(host_mirror=None)
        sync()
        dp_ms = (time.perf_counter() - t1) / k * 1e3
        extra['device_lens']['produced_on_device'] = {
            'call': 'C(data, lens_on_device.clone()).pack() -> reduce_sum', 'ms_per_step': round(dp_ms, 4),
            'value': round(N * H / (dp_ms * 1e-3) / 1e6, 1),
            'frac_of_hbm_peak_wall': round((3.0 * N * H * e + 1.0 * B * H * e + 8.0 * (4 * B + T)) / (dp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # extension, reported beside the graded pipeline: pack + reduce fused into one pass (same outputs)
        p_ref = p
        del p
        for _ in range(2):      # allocator warm-up for the fused variant's buffers
            pf, of = ta.pack_reduce(ta.with_host_sizes(data, lens_host), 'sum', fused=True)
        sync()
        timer.enabled = True
        t2 = time.perf_counter()
        for _ in range(k):
            pf, of = ta.pack_reduce(ta.with_host_sizes(data, lens_host), 'sum', fused=True)
        sync()
        fused_ms = (time.perf_counter() - t2) / k * 1e3
        timer.enabled = False
        # the same PackedSequence; the sums are the same fp32 accumulation (bit-identical whenever both paths give a
        # sequence to one wave, as at the north-star shape; few-but-long batches fold through a team of waves in the
        # two-kernel path, which associates the partial sums differently)
        assert torch.equal(pf.data, p_ref.data) and pf.data.shape == data.shape
        assert torch.allclose(of.float(), out.float(), rtol=2 ** -7, atol=1e-2)
        extra['fused_pack_reduce'] = {
            'ms_per_step': round(fused_ms, 4), 'value': round(N * H / (fused_ms * 1e-3) / 1e6, 1),
            'kernel_ms': round(timer.mean_ms('pack_reduce') or 0.0, 4),
            'hbm_bytes_moved': 2.0 * N * H * e + 1.0 * B * H * e,
            'note': 'one kernel returns the PackedSequence AND the [B,H] sums; 2/3 of the pipeline traffic'}
        del pf, of, p_ref
        # [r5] what a plain copy gets on THIS box, in THIS process, through the same mover: c.roll(0) streams the payload
        # from one CattedSequence into another (no gather).  Boxes of the pool differ by +-4 % and buffer placement by
        # as much again; the pack kernel's rate over this one is the figure that does not move with them
        c_stream = ta.with_host_sizes(data, lens_host)
        cr = c_stream.roll(0)
        sync()
        evs = []
        for _ in range(k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cr = c_stream.roll(0)
            e1.record()
            evs.append((e0, e1))
        sync()
        copy_ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
        del cr
        extra['copy_ceiling'] = {'op': 'c.roll(0): a streaming copy of the same payload through the row mover, same process',
                                 'ms': round(copy_ms, 4), 'GBps': round(2.0 * N * H * e / (copy_ms * 1e-3) / 1e9, 1),
                                 'frac_of_hbm_peak': round(2.0 * N * H * e / (copy_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # [r5] forward + backward: pack -> reduce_sum -> backward to `data` (a training step's share of this path).
        # Algorithmic bytes: the forward's 3 N H e + B H e, the reduce's backward writes N H e (and reads the [B, H]
        # cotangent), the pack's backward (P -> C, the adjoint move) reads and writes N H e: 6 N H e + 2 B H e
        xg = data.detach().requires_grad_(True)
        cot = torch.ones((B, H), dtype=data.dtype, device=dev)

        def fwd_bwd():
            out_ = ta.reduce_sum(ta.with_host_sizes(xg, lens_host).pack())
            return torch.autograd.grad(out_, xg, cot)[0]

        for _ in range(2):
            gx = fwd_bwd()
        sync()
        t3 = time.perf_counter()
        for _ in range(k):
            gx = fwd_bwd()
        sync()
        fb_ms = (time.perf_counter() - t3) / k * 1e3
        fb_bytes = 6.0 * N * H * e + 2.0 * B * H * e
        assert gx.shape == data.shape
        del gx
        extra['fwd_bwd'] = {
            'call': 'g = autograd.grad(reduce_sum(C(x, lens).pack()), x, ones): the timed pipeline plus its backward '
                    '(rua_segment_reduce_backward over the PackedSequence, then the adjoint move P -> C)',
            'ms_per_step': round(fb_ms, 4), 'steps': k, 'value': round(N * H / (fb_ms * 1e-3) / 1e6, 1), 'unit': 'M elements/s',
            'algorithmic_bytes': fb_bytes,
            'frac_of_hbm_peak_wall': round(fb_bytes / (fb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


    # the host side of a rank, with every other rank of the node live: the enqueue loop (Python + ctypes per step) and
    # the host sort of this rank's lengths, all ranks sorting at the same moment (per-rank host time
    # must stay under the per-rank kernel time when eight ranks share one host)
    from torchrua_amd import _meta
    if use_dist:
        dist.barrier()
    ts = time.perf_counter()
    for _ in range(20):
        _meta.host_sort_desc(lens_host)
    sort_ms = (time.perf_counter() - ts) / 20 * 1e3
    host_enqueue_ms = sum(host_ms) / max(1, len(host_ms))
    cpu_sets = [list(my_cpus)]

    # per-rank facts: rows, own wall clock, own kernel times (HIP events on the launch stream), own host times
    mine = [float(N), dt, timer.mean_ms('to_pack') or 0.0, timer.mean_ms('reduce') or 0.0, host_enqueue_ms, sort_ms,
            float(torch.get_num_threads())]
    per_rank = [mine]
    ranks_reported = 1
    if use_dist:
        ranks_reported = dist.get_world_size()
        if ranks_reported != args.gpus:
            raise SystemExit(f'bench.py: the process group reports {ranks_reported} ranks, --gpus {args.gpus}')
        table = torch.zeros((world, len(mine)), dtype=torch.float64, device=dev)
        table[rank] = torch.tensor(mine, dtype=torch.float64, device=dev)
        dist.all_reduce(table)                      # every rank fills its own row
        per_rank = table.cpu().tolist()
        cpu_sets = [None] * world
        dist.all_gather_object(cpu_sets, list(my_cpus))
        dt = max(row[1] for row in per_rank)        # the slowest rank sets the job's time
    n_total = sum(row[0] for row in per_rank)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = n_total * H / (dt / args.steps) / 1e6
        move_ms, red_ms = timer.mean_ms('to_pack'), timer.mean_ms('reduce')
        pack_bytes = 2.0 * N * H * e + 8.0 * (3 * B + T)
        reduce_bytes = 1.0 * N * H * e + 1.0 * B * H * e + 8.0 * B
        achieved = pack_bytes / (move_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters are NOT measured by this run (rocprofv3 --pmc needs its own
        # passes): the figure is carried from the committed profile of this same command and shape, with its source
        # named beside it, and is null for any other shape
        traffic, traffic_source = None, None
        import glob
        tpaths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic.json')))      # the latest round's
        if tpaths and (B, H, args.lo, args.hi) == (65536, 512, 8, 512):
            with open(tpaths[-1]) as f:
                tj = json.load(f)
            traffic = tj.get('to_pack_hbm_bytes_per_launch')
            traffic_source = (f"profiles/{os.path.basename(tpaths[-1])} "
                              f"({tj.get('collected', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes')})")
        line = {
            'metric': f'pack->reduce throughput, {B} seqs/GPU h={H} bf16 (M elements/s) + % HBM roofline',
            'value': round(value, 1), 'unit': 'M elements/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'pack->reduce_sum: {B} seqs/GPU, len~U({args.lo},{args.hi}), hidden={H}, bf16 '
                                   f'({"north-star shape" if (B, H, args.lo, args.hi) == (65536, 512, 8, 512) else "custom shape"}; N={N} rows on rank 0)',
                       'sharding': f'{world} x contiguous batch shards ({B * world} sequences in all), one all-gather of [B,H]' if world > 1 else 'none',
                       'lens_source': 'host (C.new-style hand-over)',
                       'ranks_reported_by_process_group': ranks_reported,
                       'cpus_of_rank0': len(my_cpus), 'torch_threads_rank0': torch.get_num_threads(),
                       'numa_node_of_rank0_gpu': _gpu_numa_node(dev),
                       'host_sort': os.environ.get('RUA_HOST_SORT', 'self-tuned'),
                       'host_sort_threads_rank0': _meta.host_sort_threads(),
                       'host_sort_native': _meta.host_sort_is_native(),      # False: the self-test kept torch.sort
                       'backend': (os.environ.get('RUA_BENCH_BACKEND', 'nccl') + (' (RCCL)' if os.environ.get('RUA_BENCH_BACKEND', 'nccl') == 'nccl' else '')) if use_dist else None},
            'per_rank': [{'rank': r, 'rows': int(row[0]), 'ms_per_step': round(row[1] / args.steps * 1e3, 4),
                          'pack_kernel_GBps': round((2.0 * row[0] * H * e + 8.0 * (3 * B + T)) / (row[2] * 1e-3) / 1e9, 1) if row[2] else None,
                          'reduce_kernel_GBps': round((row[0] * H * e + 1.0 * B * H * e + 8.0 * B) / (row[3] * 1e-3) / 1e9, 1) if row[3] else None,
                          'kernel_ms': round(row[2] + row[3], 4), 'host_enqueue_ms': round(row[4], 4),
                          'host_sort_ms': round(row[5], 4), 'torch_threads': int(row[6]),
                          'cpus': _ranges(cpu_sets[r]) if r < len(cpu_sets) and cpu_sets[r] is not None else None}
                         for r, row in enumerate(per_rank)],
            'roofline': {'bound': 'hbm', 'kernel': 'move_rows_kernel<16,false,NT,1,16,256,4> (C->P pack, 16-row tiles)',
                         'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                         'algorithmic_bytes': pack_bytes, 'avg_ms': round(move_ms, 4)},
            'reduce_kernel': {'kernel': 'seg_reduce_kernel<bf16,8,SUM,NT> (over P)', 'avg_ms': round(red_ms, 4),
                              'achieved': round(reduce_bytes / (red_ms * 1e-3) / 1e9, 1), 'unit': 'GB/s',
                              'frac': round(reduce_bytes / (red_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            'pipeline': {'algorithmic_bytes': pack_bytes + reduce_bytes,
                         'kernel_ms': round(move_ms + red_ms, 4),
                         'frac_of_hbm_peak_kernels': round((pack_bytes + reduce_bytes) / ((move_ms + red_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         'frac_of_hbm_peak_wall': round((pack_bytes + reduce_bytes) * world / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4)},
        }
        line.update(extra)
        if 'copy_ceiling' in extra:
            line['roofline']['frac_of_copy_ceiling'] = round(achieved / extra['copy_ceiling']['GBps'], 4)
            line['roofline']['copy_ceiling_GBps'] = extra['copy_ceiling']['GBps']
        if world == 1 and not args.no_cpu_baseline:
            line['parity'] = parity_leg(ta, data, lens_host, out_graded)
            line['cpu_baseline'] = cpu_baseline(args)
            line['cpu_baseline_torch'] = cpu_baseline_torch(args)
        print(json.dumps(line), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
