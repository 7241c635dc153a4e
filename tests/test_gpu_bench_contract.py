"""bench.py keeps the driver's contract: ONE JSON line with the agreed keys (run here on a tiny shape)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '512', '--hidden', '64', '--steps', '3',
                          '--warmup', '1', '--cpu-sample', '64'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'bf16' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 'traffic' in r
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    ct = d['cpu_baseline_torch']                      # the stock-torch leg, cores stated
    assert ct['kind'] == 'stock torch' and ct['cores'] >= 1 and ct['value'] > 0 and 'pack_sequence' in ct['sample']
    dl = d['device_lens']                             # the reference-signature call, first class with its own fraction
    assert dl['value'] > 0 and 0 < dl['frac_of_hbm_peak_wall'] < 1 and d['value_device_lens'] == dl['value']
    assert 'traffic_source' in r and d['config']['ranks_reported_by_process_group'] == 1
    # [r5] self-normalising and self-describing: the same-process copy ceiling, the sort in use, the reference-signature
    # number at top level, and a forward + backward leg with its own bytes
    assert d['value_reference_signature'] == dl['value'] and isinstance(d['config']['host_sort_native'], bool)
    cc = d['copy_ceiling']
    assert cc['ms'] > 0 and abs(r['frac_of_copy_ceiling'] - r['achieved'] / cc['GBps']) < 1e-3
    fb = d['fwd_bwd']
    assert fb['ms_per_step'] > 0 and fb['value'] > 0 and fb['algorithmic_bytes'] > d['pipeline']['algorithmic_bytes']
    assert d['value'] > 0 and d['ms_per_step'] > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the script itself starts two ranks (fresh children, before any GPU
    call in the parent) and rank 0 prints ONE line for a two-rank job.  Rehearsed on the one-GPU box: both ranks on
    card 0, gloo instead of RCCL (RCCL refuses two ranks on one device)."""
    env = dict(os.environ, RUA_BENCH_DEVICE='0', RUA_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '512', '--hidden', '64',
                          '--steps', '3', '--warmup', '1'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['ranks_reported_by_process_group'] == 2
    assert [r['rank'] for r in d['per_rank']] == [0, 1]
    assert all(r['rows'] > 0 and r['pack_kernel_GBps'] > 0 and r['reduce_kernel_GBps'] > 0 for r in d['per_rank'])
    assert d['per_rank'][0]['rows'] != d['per_rank'][1]['rows']          # every rank draws its own shard
    total = sum(r['rows'] for r in d['per_rank'])
    assert abs(d['value'] - total * 64 / (d['ms_per_step'] * 1e-3) / 1e6) / d['value'] < 1e-3
    assert 'cpu_baseline' not in d                                       # rank 0 at N = 1 only


def test_bench_refuses_a_world_that_is_not_what_was_asked_for():
    """--gpus N must never end as a line for a different number of ranks: a launcher environment that disagrees is an
    error (checked before anything touches a GPU, so this runs anywhere)."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and 'WORLD_SIZE=1' in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    env['WORLD_SIZE'] = '4'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and 'WORLD_SIZE=4' in out.stderr


def test_bench_self_launch_needs_the_gpus_it_was_asked_for():
    """No launcher, --gpus 16: more ranks than any single node of this pool has cards — a loud non-zero exit, never a
    one-rank run that prints n_gpus: 1 (what round 1 did)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RUA_BENCH_DEVICE')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '16', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and '--gpus 16' in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]


@pytest.mark.gpu
def test_bench_legs_are_sane_at_the_north_star_shape():
    """keys are not enough — a 150 ms fused leg around a 6 ms kernel passed the key check.  At the
    north-star shape (the driver's own command line, fewer steps) every leg's wall clock per step must stay within
    1.25x of the kernels it enqueues, the reference-signature leg (blocking read-back per step) within 1.4x, and the
    achieved float error must be on the line."""
    import gc
    import torch
    # the child process needs the memory and THIS process holds it: the full-size tests before this one leave hundreds
    # of GB in the caching allocator (r3: the test skipped itself in every full-suite run)
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < 80 << 30:
        pytest.skip('needs ~75 GB of free HBM')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '3', '--cpu-sample', '256'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert 'north-star shape' in d['config']['workload']
    kernels = d['pipeline']['kernel_ms']
    assert kernels > 0 and d['ms_per_step'] <= 1.25 * kernels, (d['ms_per_step'], kernels)
    f = d['fused_pack_reduce']
    assert f['kernel_ms'] > 0 and f['ms_per_step'] <= 1.25 * f['kernel_ms'], f
    assert d['device_lens']['ms_per_step'] <= 1.4 * kernels, (d['device_lens'], kernels)
    on_dev = d['device_lens']['produced_on_device']            # lengths a previous kernel left on the device
    assert on_dev['ms_per_step'] <= 1.4 * kernels and 0.6 <= on_dev['frac_of_hbm_peak_wall'] < 1
    assert d['roofline']['frac'] >= 0.6 and d['pipeline']['frac_of_hbm_peak_wall'] >= 0.6
    assert 0.85 <= d['roofline']['frac_of_copy_ceiling'] <= 1.1, d['roofline']      # the gather costs a few % of a plain copy
    assert d['config']['host_sort_native'] is True
    fb = d['fwd_bwd']                                          # forward + backward: twice the passes, the same rate class
    assert fb['frac_of_hbm_peak_wall'] >= 0.55 and fb['ms_per_step'] <= 2.6 * d['ms_per_step'], fb
    par = d['parity']
    assert par['max_exact'] is True and par['sum_f32_vs_fp64_over_sum_abs'] <= 1e-5
    assert par['logsumexp_f32_vs_reference_max_rel'] <= 1e-5 and par['pipeline_bf16_sum_vs_exact_rounded_to_bf16_max_ulps'] <= 1
