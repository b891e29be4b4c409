import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.hostcpu import limit_host_threads
    limit_host_threads()  # the GPU boxes report 256 cores under a 16-core cgroup quota


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
