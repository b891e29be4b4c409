"""GPU box: the segment-level recurrence, forward and backward, launch-per-step path against the persistent launches
(csrc/seg_persist.hip) at the BASELINE small-batch shapes. Prints ms per pass and us per time step.
    python tools/seg_persist_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd import kernels  # noqa: E402
from tests.test_kernels_gpu import _seg_params, rnd  # noqa: E402

K = kernels.get_kernels()
os.environ['TWOG_PERSIST_CHECK'] = 'lazy'   # time the launches, not the read-back of their error words
DEV = 'cuda:0'
for (bs, T, H, O, h) in [(8, 120, 2, 4, 512), (16, 120, 2, 9, 64), (1, 120, 1, 5, 512)]:
    pg = _seg_params(DEV, bs, T, H, O, h, (True, True, True, True), True)
    dh_h, dh_o = rnd(bs, T, H, 2 * h, seed=31).to(DEV), rnd(bs, T, O, 2 * h, seed=32).to(DEV)
    for mode in ('0', 'auto'):
        os.environ['TWOG_SEG_PERSIST'] = mode
        for what in ('fwd', 'bwd'):
            bufs = K.segrnn_fwd(pg)
            fn = (lambda: K.segrnn_fwd(pg)) if what == 'fwd' else (lambda: K.segrnn_bwd(pg, bufs, dh_h, dh_o))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.time()
            n = 10
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            used = K.last_segrnn_persistent if what == 'fwd' else K.last_segrnn_bwd_persistent
            print(f'bs={bs} T={T} H={H} O={O} h={h} {what} persistent={used}: {dt * 1e3:.3f} ms per pass, '
                  f'{dt / T * 1e6:.1f} us per step', flush=True)
        kernels.HipKernels._lazy.clear()
