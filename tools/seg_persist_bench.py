# quick timing of the segment forward: persistent vs per-step at c2 / c5 shapes
import os, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import twog_gcn_amd
from twog_gcn_amd import kernels
from tests.test_kernels_gpu import _seg_params
K = kernels.get_kernels()
os.environ['TWOG_PERSIST_CHECK'] = 'lazy'
for (bs, T, H, O, h) in [(8, 120, 2, 4, 512), (16, 120, 2, 9, 64), (1, 120, 1, 5, 512)]:
    pg = _seg_params('cuda:0', bs, T, H, O, h, (True, True, True, True), True)
    for mode in ('0', 'auto'):
        os.environ['TWOG_SEG_PERSIST'] = mode
        for _ in range(3):
            K.segrnn_fwd(pg)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 10
        for _ in range(n):
            K.segrnn_fwd(pg)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        print(f'shape bs={bs} T={T} H={H} O={O} h={h} persist={mode} used={K.last_segrnn_persistent}: {dt*1e3:.3f} ms per pass, {dt/T*1e6:.1f} us per step', flush=True)
    kernels.HipKernels._lazy.clear()
