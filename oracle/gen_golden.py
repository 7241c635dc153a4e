"""Generate tests/golden/* by running the REFERENCE itself (speedcell4/torchrua 0.5.1, imported
read-only from /root/reference, CPU).  Only inputs and the reference's outputs are stored — data,
never reference source.  Run here (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Outputs:
    tests/golden/small.npz     every hot-path function on tiny + seeded random cases
    tests/golden/sha.json      SHA-256 of the reference's outputs on reduced BASELINE.json configs
"""
import hashlib
import json
import os
import sys

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)

import torchrua as ref  # noqa: E402  (the reference)
from torchrua import C, L, P, R  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
KINDS = {'C': C, 'L': L, 'P': P, 'R': R}
FILL = -1.5

store = {}
skipped = []


def npy(t):
    t = t.detach().cpu()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def put(case, name, value):
    store[f'{case}/{name}'] = npy(value) if isinstance(value, torch.Tensor) else np.asarray(value)


def put_seq(case, name, z):
    if isinstance(z, P):
        put(case, f'{name}.data', z.data)
        put(case, f'{name}.batch_sizes', z.batch_sizes)
        put(case, f'{name}.sorted_indices', z.sorted_indices)
        put(case, f'{name}.unsorted_indices', z.unsorted_indices)
    else:
        put(case, f'{name}.data', z.data)
        put(case, f'{name}.token_sizes', z.token_sizes)


def layout_case(case, lens, H, dtype, seed):
    """All layout / select functions for one batch."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.as_tensor(lens, dtype=torch.long)
    N = int(lens.sum())
    shape = (N,) if H == 0 else (N, H)
    if dtype == torch.long:
        data = torch.randint(-1000, 1000, shape, generator=g)
    else:
        data = torch.randn(shape, generator=g).to(dtype)
    put(case, 'lens', lens)
    put(case, 'data', data)
    put(case, 'fill', np.float64(FILL))
    c = C(data=data, token_sizes=lens)
    fill = FILL if dtype != torch.long else -7
    seqs = {'C': c, 'L': c.left(fill), 'P': c.pack(), 'R': c.right(fill)}
    put(case, 'sorted_indices', seqs['P'].sorted_indices)
    for k, z in seqs.items():
        put_seq(case, f'new.{k}', z)
        b_ptr, t_ptr = z.ptr()
        put(case, f'ptr.{k}.batch', b_ptr)
        put(case, f'ptr.{k}.token', t_ptr)
        put(case, f'idx.{k}', z.idx().data)
        put(case, f'offsets.{k}', z.offsets())
        put(case, f'size.{k}', np.asarray(z.size(), dtype=np.int64))
        put(case, f'mask.{k}', ref.get_mask(z))
        put(case, f'last.{k}', z.last())
        for dst in 'CLPR':
            if dst in 'LR':
                out = getattr(z, {'L': 'left', 'R': 'right'}[dst])(fill)
            else:
                out = getattr(z, {'C': 'cat', 'P': 'pack'}[dst])()
            put_seq(case, f'cast.{k}.{dst}', out)
        T = int(lens.max())
        m = int(lens.min())
        for s in sorted({-T - 1, -1, 0, 1, 2, T, T + 1}):
            put_seq(case, f'roll.{k}.{s}', z.roll(s))
        put_seq(case, f'rev.{k}', z.rev())
        for n in sorted({1, m}):
            put_seq(case, f'head.{k}.{n}', z.head(n))
        for a, b in sorted({(0, 0), (m - 1, 0), (0, m - 1), ((m - 1) // 2, (m - 1) - (m - 1) // 2)}):
            put_seq(case, f'trunc.{k}.{a}.{b}', z.trunc((a, b)))
    # masks with values (mask.py:6-38)
    put(case, 'bmask', seqs['C'].bmask())
    if dtype.is_floating_point:
        put(case, 'fmask', seqs['L'].fmask())
    put(case, 'mask.long', seqs['P'].mask(zero=-1, one=2, dtype=torch.long))
    # tuple-key getitem / setitem (core/get.py, core/set.py)
    M = max(1, N // 2)
    pick = torch.randperm(N, generator=g)[:M]
    b_all, t_all = c.ptr()
    bp, tp = b_all[pick], t_all[pick]
    put(case, 'key.batch', bp)
    put(case, 'key.token', tp)
    value = (torch.randn((M,) + tuple(data.shape[1:]), generator=g).to(dtype) if dtype != torch.long
             else torch.randint(-50, 50, (M,) + tuple(data.shape[1:]), generator=g))
    put(case, 'key.value', value)
    for k, z in seqs.items():
        put(case, f'getitem.{k}', z[bp, tp])
        if k == 'P':
            z2 = z._replace(data=z.data.clone())
        else:
            z2 = z._replace(data=z.data.clone())
        z2[bp, tp] = value
        put(case, f'setitem.{k}', z2.data)


def reduce_case(case, lens, H, seed, zero_len=False, nan_at=None):
    g = torch.Generator().manual_seed(seed)
    lens = torch.as_tensor(lens, dtype=torch.long)
    N = int(lens.sum())
    data = torch.randn((N, H), generator=g)
    if nan_at is not None:
        data[nan_at] = float('nan')
    put(case, 'lens', lens)
    put(case, 'data', data)
    for name in ('max', 'min', 'sum', 'mean', 'prod', 'logsumexp', 'head', 'last'):
        if zero_len and name in ('head', 'last'):
            continue  # the reference's head/last need len >= 1
        fn = getattr(ref, f'segment_{name}')
        put(case, f'segment_{name}', fn(data, lens))
    # scatter_*: random permutation of rows, both include_self values (tests/test_reduce.py:26-35)
    S = lens.numel()
    index = torch.repeat_interleave(torch.arange(S), lens)
    perm = torch.randperm(N, generator=g)
    tensor = torch.randn((S, H), generator=g)
    put(case, 'scatter.index', index[perm])
    put(case, 'scatter.source', data[perm])
    put(case, 'scatter.tensor', tensor)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for name in ('max', 'min', 'sum', 'mean', 'prod', 'logsumexp'):
            fn = getattr(ref, f'scatter_{name}')
            for inc in (False, True):
                put(case, f'scatter_{name}.{int(inc)}', fn(tensor, index[perm], data[perm], include_self=inc))


def seg_case(case, lens, H, seed):
    """X.seg(duration, fn) for 4 sequence layouts x 4 duration layouts (segment.py:6-50)."""
    g = torch.Generator().manual_seed(seed)
    lens = [int(x) for x in lens]
    inputs = [torch.randn((n, H), generator=g) for n in lens]
    durations = []
    for n in lens:
        # runs of length >= 1 summing to <= n (tests/test_segment.py:81-84)
        cuts = torch.unique(torch.randint(n, (n,), generator=g), return_counts=True)[1]
        durations.append(cuts)
    put(case, 'lens', np.asarray(lens, dtype=np.int64))
    put(case, 'data', torch.cat(inputs))
    put(case, 'dur.lens', np.asarray([d.numel() for d in durations], dtype=np.int64))
    put(case, 'dur.data', torch.cat(durations))
    put(case, 'sorted_indices', P.new(inputs).sorted_indices)
    put(case, 'dur.sorted_indices', P.new(durations).sorted_indices)
    for name in ('max', 'sum', 'mean', 'logsumexp', 'last', 'head', 'min', 'prod'):
        fn = getattr(ref, f'segment_{name}')
        for ks in 'CLPR':
            for kd in 'CLPR':
                if name not in ('max', 'sum') and ks != kd:
                    continue  # full 4x4 grid for two reducers, diagonal for the rest
                try:
                    out = KINDS[ks].new(inputs).seg(KINDS[kd].new(durations), fn)
                except RuntimeError as e:
                    # reference limitation: segment_head/last over a padded layout hit a zero-length
                    # padding run on the longest sequence and torch.split_with_sizes rejects it
                    skipped.append(f'{case}/seg.{name}.{ks}.{kd}: {type(e).__name__}')
                    continue
                put_seq(case, f'seg.{name}.{ks}.{kd}', out)


def sha(t):
    a = npy(t) if isinstance(t, torch.Tensor) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_configs():
    """Reduced replicas of BASELINE.json configs; inputs are regenerated from the seed in tests
    (g = Generator().manual_seed(cfg); lens = randint; data = randn.to(dtype))."""
    out = {}

    def inputs(seed, B, lo, hi, H, dtype):
        g = torch.Generator().manual_seed(seed)
        lens = torch.randint(lo, hi + 1, (B,), generator=g)
        data = torch.randn(int(lens.sum()), H, generator=g).to(dtype)
        return lens, data

    # cfg1 (full size): cat_sequence -> pad
    lens, data = inputs(1, 32, 4, 64, 32, torch.float32)
    c = C(data, lens)
    l = c.left(0)
    out['cfg1'] = dict(seed=1, B=32, lo=4, hi=64, H=32, dtype='float32',
                       left_data=sha(l.data), left_sizes=sha(l.token_sizes), right_data=sha(c.right(0).data))
    # cfg2 / 16: pack + reduce (sum over sequences; integer outputs + payload hashed)
    lens, data = inputs(2, 256, 8, 512, 256, torch.bfloat16)
    c = C(data, lens)
    p = c.pack()
    out['cfg2'] = dict(seed=2, B=256, lo=8, hi=512, H=256, dtype='bfloat16',
                       pack_data=sha(p.data), batch_sizes=sha(p.batch_sizes), sorted_indices=sha(p.sorted_indices),
                       unsorted_indices=sha(p.unsorted_indices), cat_back=sha(p.cat().data),
                       ptr_batch=sha(p.ptr()[0]), ptr_token=sha(p.ptr()[1]))
    # cfg3 / 16: segment max over a CattedSequence (exact in any dtype)
    lens, data = inputs(3, 1024, 1, 64, 512, torch.bfloat16)
    out['cfg3'] = dict(seed=3, B=1024, lo=1, hi=64, H=512, dtype='bfloat16',
                       segment_max=sha(ref.segment_max(data, lens)), segment_min=sha(ref.segment_min(data, lens)))
    # cfg4 / 256: roll + head + last on a PackedSequence
    lens, data = inputs(4, 256, 16, 1024, 64, torch.bfloat16)
    p = C(data, lens).pack()
    out['cfg4'] = dict(seed=4, B=256, lo=16, hi=1024, H=64, dtype='bfloat16',
                       roll1=sha(p.roll(1).data), roll_neg3=sha(p.roll(-3).data), last=sha(p.last()),
                       head16_data=sha(p.head(16).data), head16_batch_sizes=sha(p.head(16).batch_sizes),
                       sorted_indices=sha(p.sorted_indices))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    # (i) the hand-checkable case of SURVEY.md §8c
    lens = [2, 4, 1, 3]
    data = torch.tensor([10, 11, 20, 21, 22, 23, 30, 40, 41, 42], dtype=torch.float32)
    put('hand', 'lens', np.asarray(lens, dtype=np.int64))
    put('hand', 'data', data)
    c = C(data, torch.tensor(lens))
    put_seq('hand', 'pack', c.pack())
    put_seq('hand', 'left', c.left(-1))
    put_seq('hand', 'right', c.right(-1))
    put('hand', 'roll1', c.roll(1).data)
    put('hand', 'roll-5', c.roll(-5).data)
    put('hand', 'last', c.last())
    put('hand', 'head1', c.head(1).data)
    dur = C.new([torch.tensor([1, 1]), torch.tensor([3, 1]), torch.tensor([1]), torch.tensor([2, 1])])
    put_seq('hand', 'segmax', c.seg(dur, ref.segment_max))
    put_seq('hand', 'left_segsum', c.left(0).seg(dur, ref.segment_sum))

    # (ii) seeded random cases; B = 17.. exercises the unstable host sort, tie-heavy lens
    rng = np.random.RandomState(0)
    cases = [
        ('b1', [5], 3, torch.float32),
        ('b1t1', [1], 1, torch.float32),
        ('ties16', rng.randint(1, 4, 16), 7, torch.float32),
        ('ties17', rng.randint(1, 4, 17), 4, torch.float32),
        ('b64', rng.randint(1, 20, 64), 8, torch.float32),
        ('b200', rng.randint(1, 6, 200), 2, torch.float32),
        ('vec', rng.randint(1, 9, 23), 0, torch.float32),          # 1-D payload
        ('bf16', rng.randint(1, 12, 33), 32, torch.bfloat16),
        ('f16', rng.randint(1, 12, 19), 5, torch.float16),
        ('i64', rng.randint(1, 12, 21), 3, torch.long),
        ('f64', rng.randint(2, 12, 18), 3, torch.float64),
    ]
    for i, (name, lens, H, dtype) in enumerate(cases):
        layout_case(f'layout.{name}', lens, H, dtype, seed=100 + i)

    reduce_case('reduce.small', rng.randint(1, 6, 12), 5, seed=200)
    reduce_case('reduce.h1', rng.randint(1, 30, 40), 1, seed=201)
    reduce_case('reduce.wide', rng.randint(1, 70, 9), 40, seed=202)
    zl = rng.randint(0, 4, 30)
    zl[0] = 3
    reduce_case('reduce.zero_len', zl, 6, seed=203, zero_len=True)

    reduce_case('reduce.nan', rng.randint(1, 5, 6), 4, seed=204, nan_at=(3, 2))

    seg_case('seg.a', rng.randint(1, 9, 7), 3, seed=300)
    seg_case('seg.b', rng.randint(1, 25, 20), 5, seed=301)

    np.savez_compressed(os.path.join(OUT, 'small.npz'), **store)
    with open(os.path.join(OUT, 'sha.json'), 'w') as f:
        json.dump(sha_configs(), f, indent=1, sort_keys=True)
    meta = dict(reference='speedcell4/torchrua 0.5.1 (/root/reference)', torch=torch.__version__,
                numpy=np.__version__, n_arrays=len(store), reference_raised=skipped)
    with open(os.path.join(OUT, 'META.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(store), 'arrays;', os.path.getsize(os.path.join(OUT, 'small.npz')), 'bytes')


def extra():
    """Round-2 additions (tests/golden/extra.npz; small.npz is left byte-for-byte as it was): the sizes SURVEY.md
    §8c(ii) asked for beyond B = 200 — a tie-heavy batch of 1 000 sequences, 512-wide reductions, sequences of
    512-1 024 rows — and a batch with zero-length sequences through pack()."""
    global store
    store = {}
    rng = np.random.RandomState(20)
    layout_case('layout.b1000ties', rng.randint(1, 5, 1000), 2, torch.float32, seed=400)
    layout_case('layout.b120wide', rng.randint(1, 40, 120), 1, torch.float32, seed=401)
    reduce_case('reduce.h512', rng.randint(1, 32, 12), 512, seed=402)
    reduce_case('reduce.long', rng.randint(512, 1025, 6), 8, seed=403)
    zl = rng.randint(0, 3, 1000)
    zl[0] = 2
    reduce_case('reduce.b1000zero', zl, 3, seed=404, zero_len=True)
    # zero-length sequences through pack() (core/cast.py:41-49 handles them; P -> anything raises in the reference)
    for name, lens, H in (('empty.a', [0, 3, 0, 2], 4), ('empty.b', [2, 0, 0, 5, 1, 0], 40), ('empty.c', [0, 0, 4], 1)):
        g = torch.Generator().manual_seed(len(lens))
        data = torch.randn((sum(lens), H), generator=g)
        c = C(data, torch.tensor(lens))
        put(name, 'lens', np.asarray(lens, dtype=np.int64))
        put(name, 'data', data)
        put_seq(name, 'pack', c.pack())
        put_seq(name, 'left', c.left(FILL))
        put_seq(name, 'right', c.right(FILL))
        put_seq(name, 'left.pack', c.left(FILL).pack())
        put_seq(name, 'right.pack', c.right(FILL).pack())
        for op in ('max', 'sum', 'logsumexp'):
            put(name, f'segment_{op}', getattr(ref, f'segment_{op}')(data, torch.tensor(lens)))
    np.savez_compressed(os.path.join(OUT, 'extra.npz'), **store)
    meta_path = os.path.join(OUT, 'META.json')
    meta = json.load(open(meta_path))
    meta['extra_n_arrays'] = len(store)
    meta['extra_reference_raised'] = [x for x in skipped if x.split('/')[0] in ('layout.b1000ties', 'layout.b120wide')]
    with open(meta_path, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(store), 'extra arrays;', os.path.getsize(os.path.join(OUT, 'extra.npz')), 'bytes')


def foreign():
    """tests/golden/foreign.npz: PackedSequences whose ties are NOT in torch.sort's order (the stable order, built by
    hand) through the functions of a PackedSequence.  roll / rev end in `.pack()` in the reference (select/roll.py:26-30,
    select/rev.py:33-34), so their results come back in the host sort's order whatever the input's was."""
    global store
    store = {}
    rng = np.random.RandomState(31)
    n_differ = 0
    for name, lens, H in (('foreign.a', rng.randint(1, 5, 40), 3), ('foreign.b', rng.randint(1, 7, 300), 1),
                          ('foreign.c', rng.randint(2, 4, 64), 40)):
        g = torch.Generator().manual_seed(len(lens))
        lens = torch.as_tensor(lens, dtype=torch.long)
        data = torch.randn((int(lens.sum()), H), generator=g)
        c = C(data, lens)
        theirs = c.pack()
        stable = torch.sort(lens, descending=True, stable=True)[1]
        n_differ += int(not torch.equal(stable, theirs.sorted_indices))
        off = torch.cumsum(lens, 0) - lens
        rows = [int(off[b]) + t for t, n in enumerate(theirs.batch_sizes.tolist()) for b in stable[:n].tolist()]
        mine = P(data=data[torch.tensor(rows)], batch_sizes=theirs.batch_sizes, sorted_indices=stable,
                 unsorted_indices=ref.invert_permutation(stable))
        assert torch.equal(mine.cat().data, data)
        put(name, 'lens', lens)
        put(name, 'data', data)
        put_seq(name, 'pack', mine)
        for s_ in (-3, -1, 0, 1, 2, 7):
            put_seq(name, f'roll.{s_}', mine.roll(s_))
        put_seq(name, 'rev', mine.rev())
        put_seq(name, 'cat', mine.cat())
        put_seq(name, 'left', mine.left(FILL))
        put_seq(name, 'right', mine.right(FILL))
        put(name, 'last', mine.last())
        put_seq(name, 'head.1', mine.head(1))
        put_seq(name, 'trunc.1.0', mine.trunc((1, 0))) if int(lens.min()) > 1 else None
        put(name, 'segment_sum.via_cat', ref.segment_sum(*mine.cat()))
    assert n_differ == 3, 'every case must differ from the host sort order'
    np.savez_compressed(os.path.join(OUT, 'foreign.npz'), **store)
    meta_path = os.path.join(OUT, 'META.json')
    meta = json.load(open(meta_path))
    meta['foreign_n_arrays'] = len(store)
    with open(meta_path, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(store), 'foreign-order arrays;', os.path.getsize(os.path.join(OUT, 'foreign.npz')), 'bytes')


if __name__ == '__main__':
    if '--foreign' in sys.argv:
        foreign()
    elif '--extra' in sys.argv:
        extra()
    else:
        main()
        extra()
        foreign()
