"""Which kernels run in forward + backward of indexing, casts, selects and reductions?  Run under

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2_aten -o runc -- python3 scripts/aten_free_trace.py

and condense with `python scripts/aten_free_trace.py --summarize` -> profiles/r02_kernels_fwd_bwd.txt.
Inputs are drawn on the host and uploaded (copies, not kernels); cotangents are handed to torch.autograd.grad, so the
only ATen kernels left are the ones the product path itself launches."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize():
    f = glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof2_aten', '**', '*kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ours = [r for r in rows if 'rua::' in r['Name']]
    other = [r for r in rows if 'rua::' not in r['Name']]
    out = ['# kernels launched by scripts/aten_free_trace.py (rocprofv3 --kernel-trace --stats): forward + backward of',
           '# getitem / setitem (tuple, tensor and Z keys), scatter_* x include_self, every cast, roll/rev/last/head/trunc,',
           '# reduce_* over all four layouts.  "other" = everything that is not a rua:: kernel.', '',
           f'rua:: kernels: {len(ours)} distinct, {sum(int(r["Calls"]) for r in ours)} launches',
           f'other kernels: {len(other)} distinct, {sum(int(r["Calls"]) for r in other)} launches', '', 'other:']
    for r in other:
        out.append(f'  {int(r["Calls"]):5d} x  avg {float(r["AverageNs"]) / 1e3:8.1f} us   {r["Name"][:150]}')
    out += ['', 'rua:']
    for r in ours:
        out.append(f'  {int(r["Calls"]):5d} x  avg {float(r["AverageNs"]) / 1e3:8.1f} us   {r["Name"].replace("void ", "").split("(")[0][:120]}')
    text = '\n'.join(out) + '\n'
    open(os.path.join(ROOT, 'profiles', 'r02_kernels_fwd_bwd.txt'), 'w').write(text)
    print(text)


def main():
    import torch

    import torchrua_amd as ta
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(1, 40, (300,), generator=g)
    n, H = int(lens.sum()), 64
    x_host = torch.randn(n, H, generator=g)
    w_host = torch.randn(4 * n, H, generator=g)

    def up(t):
        return t.to(dev)

    W = up(w_host)
    c0 = ta.with_host_sizes(up(x_host), lens)
    seqs = {'C': c0, 'L': c0.left(), 'P': c0.pack(), 'R': c0.right()}
    bp_h = torch.repeat_interleave(torch.arange(300), lens)
    tp_h = torch.cat([torch.arange(int(k)) for k in lens])
    perm = torch.randperm(n, generator=g)
    bp, tp = up(bp_h[perm]), up(tp_h[perm])
    torch.cuda.synchronize()

    def grad_of(fn, leaf):
        out = fn(leaf)
        out = out.data if not isinstance(out, torch.Tensor) else out
        torch.autograd.grad(out, leaf, W[:out.numel() // H].reshape(out.shape))

    for k, z in seqs.items():
        leaf = z.data.detach().requires_grad_()
        zz = z._replace(data=leaf)
        grad_of(lambda d: zz[bp, tp], leaf)                         # tuple key
        grad_of(lambda d: zz[zz.idx()], leaf)                       # Z key
        zz2 = z._replace(data=z.data.detach().clone() if False else z.data.detach())
        zz2[bp, tp] = W[:n]                                         # setitem, tuple key
        for name, f in (('cat', lambda q: q.cat()), ('left', lambda q: q.left()), ('pack', lambda q: q.pack()),
                        ('right', lambda q: q.right()), ('roll', lambda q: q.roll(3)), ('rev', lambda q: q.rev()),
                        ('last', lambda q: q.last()), ('head', lambda q: q.head(1)), ('trunc', lambda q: q.trunc((0, 0)))):
            grad_of(lambda d: f(zz), leaf)
        for name in ('sum', 'mean', 'max', 'min', 'prod', 'logsumexp'):
            grad_of(lambda d: getattr(ta, f'reduce_{name}')(zz), leaf)
    ta.patch_tensor_indexing()
    leaf = c0.data.detach().requires_grad_()
    grad_of(lambda d: leaf[c0.idx().roll(1)], leaf)                 # tensor[Z]
    ta.unpatch_tensor_indexing()
    S = 300
    ten_h = torch.randn(S, H, generator=g)
    for name in ('sum', 'mean', 'max', 'min', 'prod', 'logsumexp'):
        for inc in (False, True):
            t = up(ten_h).requires_grad_()
            s = c0.data.detach().requires_grad_()
            out = getattr(ta, f'scatter_{name}')(t, bp, s, include_self=inc)
            need = [s] if (name in ('sum', 'logsumexp') and not inc) else [t, s]
            torch.autograd.grad(out, need, W[:S])
    torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    if '--summarize' in sys.argv:
        summarize()
    else:
        main()
