"""CPU: the launch logic of bench.py that needs no GPU -- a --gpus / WORLD_SIZE mismatch is refused with the exact
torchrun command, and a bare `--gpus N` start becomes a launcher of N child ranks (checked by substituting the child)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], cwd=ROOT,
                       env=dict(os.environ, WORLD_SIZE='2'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert '--nproc-per-node 4' in r.stderr and 'WORLD_SIZE=2' in r.stderr


def test_bare_start_launches_n_ranks(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '3'])
    assert bench.launch_ranks(8) == 7          # the launcher exits with the children's code
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '8', '--steps', '3'] and cmd[-5].endswith('bench.py')
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
