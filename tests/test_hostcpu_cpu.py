"""CPU: the host thread budget follows the cgroup quota (hostcpu.py), not the node's core count."""
import os

import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import hostcpu


def test_effective_cpu_count_is_bounded_by_the_quota():
    n = hostcpu.effective_cpu_count()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            assert n <= max(1, int(int(quota) / int(period)))
    except OSError:
        pass


def test_limit_host_threads_caps_torch():
    before = torch.get_num_threads()
    cap = hostcpu.limit_host_threads(reserve=0, share=1)
    assert torch.get_num_threads() <= max(cap, 1)
    assert hostcpu.limit_host_threads(reserve=0, share=4) <= max(1, cap)
    torch.set_num_threads(min(before, cap))
