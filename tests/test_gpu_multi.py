"""More than one GPU on the box: the sharded pipeline over RCCL with 2 / 4 / 8 ranks.  Skips on the
one-GPU boxes of this pool; on a multi-GPU node it runs the moment the devices exist.  The ranks are fresh child
processes started with the rendezvous environment (as bench.py::self_launch does) — nothing is exec'ed from a process
that holds a GPU, and at most 8 ranks touch the cards."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _launch(world: int, B: int, H: int, extra_env=None):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RUA_HOST_SORT='2', **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'multi_rank_worker.py'), str(B), str(H)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:          # exactly the children started above
                q.kill()
            raise
    return [p.returncode for p in procs], outs


def test_worker_with_one_rccl_rank():
    """The same worker with a world of one: RCCL initialises, both all-gather forms run, the result equals the
    single-process one — what a one-GPU box can rehearse of the test below."""
    codes, outs = _launch(1, 515, 96)
    assert codes == [0], outs[0][1][-2000:]
    assert 'MULTI_RANK_OK world=1 B=515' in outs[0][0]


@pytest.mark.parametrize('world', [2, 4, 8])
def test_sharded_pipeline_over_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f'{torch.cuda.device_count()} GPU(s) on this box; needs {world}')
    for B in (world * 509 + 3, world * 512):          # ragged shards, then equal ones (one ncclAllGather, async too)
        codes, outs = _launch(world, B, 96)
        assert codes == [0] * world, '\n'.join(o[1][-1500:] for o in outs)
        assert f'MULTI_RANK_OK world={world} B={B}' in outs[0][0]


@pytest.mark.parametrize('world', [2, 8])
def test_bench_line_over_rccl(world):
    """bench.py --gpus N on a real multi-GPU node, small shape: one line, N ranks, per-rank rates."""
    import json
    if torch.cuda.device_count() < world:
        pytest.skip(f'{torch.cuda.device_count()} GPU(s) on this box; needs {world}')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RUA_BENCH_DEVICE', 'RUA_BENCH_BACKEND')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--batch', '4096', '--hidden',
                          '128', '--steps', '5', '--warmup', '2'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert d['n_gpus'] == world and d['config']['backend'].startswith('nccl') and len(d['per_rank']) == world
    assert d['config']['cpus_of_rank0'] >= 1 and d['scaling'] == 'weak'
