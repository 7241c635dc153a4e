"""HBM traffic of single operators from the PMC counters (passes over the keys of index_buckets,
over-fetch of narrow rows and pads).

    # on the GPU box, one pass per counter (the pool refuses --pmc together with trace domains):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcops_fetch -o run -- python3 scripts/pmc_ops.py run
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcops_write -o run -- python3 scripts/pmc_ops.py run
    python3 scripts/pmc_ops.py run --time > gpurun_out/pmcops_time.json          # HIP-event times, no profiler
    python3 scripts/pmc_ops.py summarize r04                                     # -> profiles/r04_pmc_ops.json

`run` executes a fixed list of operator groups; every group is one uncounted warm-up and `REPS` counted launches of one
operator, and a tiny mask launch (rua::mask_kernel<unsigned long, 1>: get_mask of a two-sequence batch, used by none of the operators measured) separates the groups, so the counter
rows — one per dispatch, in dispatch order — can be cut into groups without naming kernels.  Counter unit: KiB.
gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section), so
reads are doubled; other access widths are uncalibrated in absolute terms — the `calib.*` groups (a streaming copy
through the same mover at 1 KiB, 64 B and 32 B rows, whose traffic is known) give the scale for each width.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 3
DEFAULT = ('ns', 'w32', 'w64')      # the families of round 4's committed file; --only=bwd,odd,w8 ... picks others
ONLY = [a.split('=', 1)[1].split(',') for a in sys.argv if a.startswith('--only=')]
ONLY = ONLY[0] if ONLY else []
MARK = 'mask_kernel<unsigned long'


def groups():
    """[(name, algorithmic_bytes, callable)] — built lazily: every group allocates its own inputs and frees them."""
    import torch

    import torchrua_amd as ta
    from torchrua_amd import _ops as O
    from torchrua_amd.layout import describe
    dev = torch.device('cuda:0')

    def ragged(seed, B, lo, hi, H, dtype=torch.bfloat16):
        g = torch.Generator().manual_seed(seed)
        lens = torch.randint(lo, hi + 1, (B,), generator=g)
        n = int(lens.sum())
        data = torch.empty((n, H), dtype=dtype, device=dev)
        step = 1 << 22
        for a in range(0, n, step):
            data[a:a + step] = torch.randn((min(n, a + step) - a, H), device=dev)
        return lens, data

    def north_star():
        lens, data = ragged(5, 65536, 8, 512, 512)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        N, H, B, T, e = data.size(0), 512, 65536, int(lens.max()), 2
        bp = c.ptr()[0]
        shuffled = bp[torch.randperm(N, device=dev)]
        yield 'ns.index_buckets(shuffled)', 2 * N * 8, lambda: O.index_buckets(shuffled, B)
        yield 'ns.index_buckets(p.ptr()[0])', 2 * N * 8, (lambda ix=p.ptr()[0]: O.index_buckets(ix, B))
        ten = torch.zeros(B, H, dtype=torch.bfloat16, device=dev)
        yield 'ns.scatter_sum(shuffled)', N * H * e + B * H * e + 2 * N * 8, lambda: ta.scatter_sum(ten, shuffled, data)
        yield 'ns.c.pack()', 2 * N * H * e, lambda: c.pack()
        yield 'ns.p.cat()', 2 * N * H * e, lambda: p.cat()
        yield 'ns.c.left()', N * H * e + B * T * H * e, lambda: c.left()
        yield 'ns.p.left()', N * H * e + B * T * H * e, lambda: p.left()
        yield 'ns.c.roll(0)  [calib: streaming copy, 1 KiB rows]', 2 * N * H * e, lambda: c.roll(0)
        lft = c.left()
        yield 'ns.l.idx()', N * 8, lambda: lft.idx()
        yield 'ns.c.bmask()', B * T, lambda: c.bmask()

    def narrow(H):
        rows = int(8e9 / (H * 2))
        B = max(1024, rows // 260)
        lens, data = ragged(H, B, 8, 512, H)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        nb = data.numel() * 2
        cl, pl = describe(c), describe(p)
        out = torch.empty_like(data)
        tag = f'w{H * 2}'
        yield f'{tag}.pack', 2 * nb, lambda: O.launch_move(O.MovePlan(pl, cl, data.shape), data, out=out)
        yield f'{tag}.P.cat', 2 * nb, lambda: O.launch_move(O.MovePlan(cl, pl, data.shape), p.data, out=out)
        yield f'{tag}.P.roll(1)', 2 * nb, lambda: O.launch_move(O.MovePlan(pl, pl, data.shape, tmap=1, arg=1), p.data, out=out)
        yield f'{tag}.C.roll(0)  [calib: streaming copy, {H * 2} B rows]', 2 * nb, lambda: O.launch_move(O.MovePlan(cl, cl, data.shape, tmap=1, arg=0), data, out=out)

    def backward():
        """[r5] the fused reduce backward (rua_segment_reduce_backward) at the north-star shape over C and over P.
        Algorithmic bytes: sum writes the gradient and reads the [B, H] cotangent; max / logsumexp also read the payload
        and out (max: and the [B, H] fp32 tie counts the forward left)."""
        lens, data = ragged(5, 65536, 8, 512, 512)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        N, H, B, e = data.size(0), 512, 65536, 2
        for name, alg in (('sum', N * H * e + B * H * e), ('max', 2 * N * H * e + B * H * (2 * e + 4)),
                          ('logsumexp', 2 * N * H * e + 2 * B * H * e)):
            for tag, z in (('C', c), ('P', p)):
                x = z.data.detach().requires_grad_(True)
                out = getattr(ta, f'reduce_{name}')(z._replace(data=x))
                cot = torch.randn_like(out)
                yield f'bwd.{name}({tag})', alg, (lambda out=out, x=x, cot=cot: torch.autograd.grad(out, x, cot, retain_graph=True))
                del x, out, cot

    def odd(H):
        """[r5] rows that are not a multiple of a 128-byte line: 8 (mod 16) bytes (H = 500 in bf16), whole vectors
        (H = 1 000, 1 080), 8 GB payloads."""
        rows = int(8e9 / (H * 2))
        B = max(1024, rows // 260)
        lens, data = ragged(H, B, 8, 512, H)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        nb = data.numel() * 2
        T = int(lens.max())
        tag = f'odd{H * 2}'
        yield f'{tag}.pack', 2 * nb, lambda: c.pack()
        yield f'{tag}.P.cat', 2 * nb, lambda: p.cat()
        yield f'{tag}.C.left', nb + B * T * H * 2, lambda: c.left()
        yield f'{tag}.C.roll(0)  [calib: streaming copy, {H * 2} B rows]', 2 * nb, lambda: c.roll(0)

    def tiny(rb):
        """[r5] 1-D payloads: rows of 1 / 2 / 4 / 8 bytes (bool masks, fp16 / fp32 scalars, int64 token ids), 500 M rows."""
        dtype = {1: torch.uint8, 2: torch.float16, 4: torch.float32, 8: torch.int64}[rb]
        B = 1923076
        g = torch.Generator().manual_seed(rb)
        lens = torch.randint(8, 513, (B,), generator=g)
        n = int(lens.sum())
        data = torch.randint(0, 100, (n,), device=dev, dtype=torch.int32).to(dtype)
        c = ta.with_host_sizes(data, lens)
        p = c.pack()
        nb = n * rb
        T = int(lens.max())
        tag = f'w{rb}'
        yield f'{tag}.pack', 2 * nb, lambda: c.pack()
        yield f'{tag}.P.cat', 2 * nb, lambda: p.cat()
        yield f'{tag}.C.left', nb + B * T * rb, lambda: c.left()
        yield f'{tag}.P.left', nb + B * T * rb, lambda: p.left()
        yield f'{tag}.P.roll(1)', 2 * nb, lambda: p.roll(1)
        yield f'{tag}.C.roll(0)  [calib: streaming copy, {rb} B rows]', 2 * nb, lambda: c.roll(0)

    import gc
    fams = [('ns', north_star)] + [(f'w{2 * H}', (lambda H=H: narrow(H))) for H in (16, 32)] + [('n16', lambda: narrow(8))]
    fams += [('bwd', backward)] + [(f'odd{2 * H}', (lambda H=H: odd(H))) for H in (500, 1000, 1080)]
    fams += [(f'w{rb}', (lambda rb=rb: tiny(rb))) for rb in (8, 4, 2, 1)]
    for fam, gen in fams:
        if ONLY and not any(fam.startswith(o) for o in ONLY):
            continue
        if not ONLY and fam not in DEFAULT:
            continue
        yield from gen()
        gc.collect()
        torch.cuda.empty_cache()


def run(timed: bool):
    import torch

    import torchrua_amd as ta
    dev = torch.device('cuda:0')
    tiny = ta.with_host_sizes(torch.zeros(4, 2, device=dev), torch.tensor([1, 3]))
    times = {}
    order = []
    for name, nbytes, fn in groups():
        ta.get_mask(tiny)                        # marker: what follows is this group's WARM-UP (not counted) ...
        fn()
        torch.cuda.synchronize()
        ta.get_mask(tiny)                        # ... marker: everything up to the next marker is this group
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ta.get_mask(tiny)                        # ... marker: what follows (the next group's set-up) is not counted
        order.append(name)
        times[name] = {'algorithmic_bytes': nbytes, 'ms_per_call': e0.elapsed_time(e1) / REPS}
    torch.cuda.synchronize()
    if timed:
        print(json.dumps({'order': order, 'ops': times}, indent=1))


def counter_rows(d, counter):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    per = collections.OrderedDict()                      # several rows per dispatch (one per instance): sum them
    for did, name, v in sorted(rows):
        if did in per:
            per[did][1] += v
        else:
            per[did] = [name, v]
    return list(per.values())


def cut(rows, order):
    """Counter rows in dispatch order -> {group: {kernel: KiB per CALL}} using the marker launches."""
    out, cur, gi = {}, None, -1
    for name, v in rows:
        if MARK in name:
            gi += 1                                  # three markers per group: warm-up | the counted launches | set-up
            cur = order[gi // 3] if gi % 3 == 1 and gi // 3 < len(order) else None
            if cur is not None:
                out[cur] = collections.defaultdict(float)
            continue
        if cur is not None:
            short = name.replace('void ', '').split('(')[0][:90]
            out[cur][short] += v / REPS
    return out


def summarize(tag, pre='pmcops'):
    G = os.path.join(ROOT, 'gpurun_out')
    meta = json.load(open(os.path.join(G, f'{pre}_time.json')))
    order = meta['order']
    fetch = cut(counter_rows(os.path.join(G, f'{pre}_fetch'), 'FETCH_SIZE'), order)
    write = cut(counter_rows(os.path.join(G, f'{pre}_write'), 'WRITE_SIZE'), order)
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of scripts/pmc_ops.py; per CALL of the operator '
                   '(all its kernels); read = 2 * FETCH_SIZE KiB (gfx950: half of a wide read stream is reported), write = '
                   'WRITE_SIZE KiB; widths other than 16 B per lane are uncalibrated in absolute terms: compare with the '
                   'calib rows of the same width', 'ops': collections.OrderedDict()}
    for name in order:
        rd = 2.0 * 1024 * sum(fetch.get(name, {}).values())
        wr = 1024.0 * sum(write.get(name, {}).values())
        alg = meta['ops'][name]['algorithmic_bytes']
        ms = meta['ops'][name]['ms_per_call']
        out['ops'][name] = {
            'algorithmic_bytes': alg, 'ms_per_call': round(ms, 4), 'algorithmic_TBps': round(alg / ms / 1e9, 3),
            'hbm_read_bytes': rd, 'hbm_write_bytes': wr, 'traffic_over_algorithmic': round((rd + wr) / alg, 3) if alg else None,
            'kernels_read_KiB_x2': {k: round(2 * v, 1) for k, v in fetch.get(name, {}).items()},
            'kernels_write_KiB': {k: round(v, 1) for k, v in write.get(name, {}).items()}}
    path = os.path.join(ROOT, 'profiles', f'{tag}.json' if pre != 'pmcops' else f'{tag}_pmc_ops.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    for name, o in out['ops'].items():
        print(f"{name:55s} {o['ms_per_call']:9.3f} ms  alg {o['algorithmic_bytes'] / 1e9:7.3f} GB  "
              f"read {o['hbm_read_bytes'] / 1e9:7.3f}  write {o['hbm_write_bytes'] / 1e9:7.3f}  x{o['traffic_over_algorithmic']}")


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'summarize':
        summarize(sys.argv[2] if len(sys.argv) > 2 else 'r04', *(sys.argv[3:4]))
    else:
        run('--time' in sys.argv)
