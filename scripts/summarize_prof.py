"""Condense rocprofv3 output under gpurun_out/ into the small, tracked files under profiles/.

    python scripts/summarize_prof.py r01          # -> profiles/r01_kernel_stats.csv, r01_pmc.json, r01_traffic.json

Expects gpurun_out/prof_stats (--kernel-trace --stats), prof_fetch (--pmc FETCH_SIZE), prof_write
(--pmc WRITE_SIZE), each from `bench.py --no-cpu-baseline` at the default (north-star) shape.
HBM bytes follow MI355X_MICROARCH.md §HBM: bytes = counter * 1024; on gfx950 FETCH_SIZE reports 1/2 of
a wide coalesced read stream, so the read side is doubled; WRITE_SIZE is exact for 16-B stores."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
PRE = os.environ.get('PROF_PREFIX', 'prof_')      # gpurun_out/<PRE>stats, <PRE>fetch, <PRE>write
P = os.path.join(ROOT, 'profiles')
os.makedirs(P, exist_ok=True)


def short(name):
    name = name.replace('void ', '')
    return name.split('(')[0][:80]


def kernel_stats():
    """Per-kernel stats from `rocprofv3 --kernel-trace --stats`: the CSV when the run wrote one, else the rocpd
    SQLite database (ROCm 7.2's default output format)."""
    files = glob.glob(os.path.join(G, PRE + 'stats', '**', '*_kernel_stats.csv'), recursive=True)
    if files:
        return [(r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs'])
                for r in csv.DictReader(open(files[0]))]
    import sqlite3
    con = sqlite3.connect(glob.glob(os.path.join(G, PRE + 'stats', '**', '*_results.db'), recursive=True)[0])
    total = con.execute('select sum(duration) from kernels').fetchone()[0]
    q = ('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels '
         'group by name order by sum(duration) desc')
    return [(n, c, t, round(a, 1), round(100.0 * t / total, 4), lo, hi) for n, c, t, a, lo, hi in con.execute(q)]


def counter_rows(kind):
    files = glob.glob(os.path.join(G, f'{PRE}{kind}', '**', '*_counter_collection.csv'), recursive=True)
    if files:
        return [(r['Kernel_Name'], r['Counter_Name'], float(r['Counter_Value'])) for r in csv.DictReader(open(files[0]))]
    dbs = glob.glob(os.path.join(G, f'{PRE}{kind}', '**', '*_results.db'), recursive=True)
    if not dbs:
        return []
    import sqlite3
    con = sqlite3.connect(dbs[0])
    # one row per (dispatch, counter instance): sum the instances of a dispatch
    q = 'select name, counter_name, sum(counter_value) from pmc_events group by dispatch_id, counter_name'
    return [(n, c, float(v)) for n, c, v in con.execute(q)]


with open(os.path.join(P, f'{tag}_kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns'])
    for name, *rest in kernel_stats():
        w.writerow([short(name)] + rest)

pmc = {}
for kind, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    for name, cname, value in counter_rows(kind):
        if cname == counter and 'rua::' in name:
            agg[short(name)].append(value)
    for k, v in agg.items():
        pmc.setdefault(k, {})[counter] = {'launches': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v)}

out = {'note': 'rocprofv3 --pmc, separate passes; counter unit KiB; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950',
       'kernels': {}}
for k, c in pmc.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        rd = 2.0 * c['FETCH_SIZE']['mean'] * 1024
        wr = c['WRITE_SIZE']['mean'] * 1024
        out['kernels'][k] = {'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                             'hbm_bytes_per_launch': rd + wr, 'raw': c}
with open(os.path.join(P, f'{tag}_pmc.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
move = [v for k, v in out['kernels'].items() if 'move_rows_kernel<16, false' in k]
if move:
    with open(os.path.join(P, f'{tag}_traffic.json'), 'w') as f:
        import datetime
        json.dump({'to_pack_hbm_bytes_per_launch': move[0]['hbm_bytes_per_launch'],
                   'collected': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --no-cpu-baseline`, '
                                f'gfx950 corrections per MI355X_MICROARCH.md, summarised {datetime.date.today().isoformat()}',
                   'source': f'profiles/{tag}_pmc.json'},
                  f, indent=1)
print(open(os.path.join(P, f'{tag}_kernel_stats.csv')).read()[:1500])
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'raw'} for k, v in out['kernels'].items()}, indent=1)[:2000])
