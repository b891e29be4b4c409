"""GPU: `bench.py --gpus N` started BARE (no torchrun, no WORLD_SIZE) must start N ranks by itself and report them.
Two ranks share the one visible MI355X here (TWOG_BENCH_BACKEND=gloo for the collectives -- RCCL needs one GPU per
rank; the driver's 8-GPU run uses the default nccl backend through exactly the same launch path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=900):
    env = dict(os.environ, TWOG_BENCH_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]   # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_weak():
    d = _bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'c5', '--no-cpu-baseline')
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak'
    assert len(d['config']['devices']) == 2 and d['config']['devices'][0].startswith('rank0:')
    assert d['config']['per_gpu_batch'] == 16 and d['config']['global_batch'] == 32
    assert d['config']['collective_backend'] == 'gloo'
    assert d['value'] > 0 and d['forward_only_clips_per_s'] > 0
    assert abs(d['value'] - 32 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']
    # the same line carries the other scaling mode: fixed GLOBAL batch of the workload (16 clips -> 8 per GPU)
    st = d['strong_scaling']
    assert st['scaling'] == 'strong' and st['per_gpu_batch'] == 8 and st['global_batch'] == 16 and st['value'] > 0
    assert 'weak scaling' in d['value_is']


def test_bench_strong_scaling_splits_the_global_batch():
    d = _bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'c5', '--scaling', 'strong',
               '--no-cpu-baseline')
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong'
    assert d['config']['per_gpu_batch'] == 8 and d['config']['global_batch'] == 16
    wk = d['weak_scaling']
    assert wk['scaling'] == 'weak' and wk['per_gpu_batch'] == 16 and wk['global_batch'] == 32 and wk['value'] > 0


def test_bench_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE' in r.stderr
