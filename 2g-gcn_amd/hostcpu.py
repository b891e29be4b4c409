"""Host CPU budget. Containers often expose every core of the node (os.cpu_count() = 256 on the MI355X boxes) while a
cgroup quota grants far fewer (16 there). Thread pools sized from cpu_count() then exhaust the quota, the kernel
throttles the whole cgroup for the rest of the 100 ms period, and the thread that feeds the GPU stalls with it (measured:
sporadic +90 ms steps). `limit_host_threads()` sizes torch's intra-op pool from the quota instead."""
import os


def effective_cpu_count() -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:  # cgroup v2
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def limit_host_threads(reserve: int = 4, share: int = 1) -> int:
    """Caps torch's CPU thread pools at this process's share (1/`share`: ranks on the node) of the cgroup quota minus
    `reserve` cores (kept for the launch thread and the HIP runtime's helper threads). Returns the cap."""
    import torch
    cap = max(1, effective_cpu_count() // max(1, share) - reserve)
    if torch.get_num_threads() > cap:
        torch.set_num_threads(cap)
    os.environ.setdefault('OMP_NUM_THREADS', str(cap))
    return cap
