"""bench.py keeps the driver's contract: ONE JSON line with the agreed keys (run here on a tiny shape)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    assert d['value'] > 0 and d['ms_per_step'] > 0
