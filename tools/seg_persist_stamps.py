"""Reads the phase stamps of the diagnostic build (tools/seg_persist_stamps.sh) of the persistent segment forward launch."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd import kernels, _lib as L  # noqa: E402
from tests.test_kernels_gpu import _seg_params  # noqa: E402

K = kernels.get_kernels()
DEV = 'cuda:0'
NAMES = {0: 'wait', 1: 'loads+products', 2: 'combine', 3: 'relu/scores/softmax', 4: 'sums or gates, stores issued', 5: 'drain+signal'}
for (bs, T, H, O, h) in [(8, 120, 2, 4, 512), (16, 120, 2, 9, 64), (1, 120, 1, 5, 512)]:
    p = _seg_params(DEV, bs, T, H, O, h, (True, True, True, True), True)
    bufs = K.segrnn_fwd(p)   # allocates the buffers (and runs once)
    s = L.SegRnn()
    K._fill_seg(s, p, bufs)
    n_sync = int(K.lib.twog_segrnn_persistent_sync_bytes()) // 4
    for rep in range(2):
        sync = torch.zeros(n_sync, dtype=torch.int32, device=DEV)
        rc = K.lib.twog_segrnn_fwd_persistent(C.byref(s), sync.data_ptr(), K._stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
    st = sync[2048:2048 + 64].cpu().view(4, 16).float() * 0.01 / T   # ticks of 10 ns -> us per step
    print(f'bs={bs} T={T} H={H} O={O} h={h}: FORWARD us per step by role and phase')
    for r, name in enumerate(('P1a', 'P1b', 'P2h', 'P2o')):
        row = ', '.join(f'{NAMES[k]} {st[r][k]:.2f}' for k in range(6) if st[r][k] > 0)
        print(f'  {name}: total {st[r][:6].sum():.2f} | {row}')
    # backward launch
    from tests.test_kernels_gpu import rnd
    dh_h, dh_o = rnd(bs, T, H, 2 * h, seed=31).to(DEV), rnd(bs, T, O, 2 * h, seed=32).to(DEV)
    out = K.segrnn_bwd(p, bufs, dh_h, dh_o)   # allocates outputs / scratch (and runs once)
    b = L.SegRnnBwd()
    b.d_hs_h, b.d_hs_o = dh_h.data_ptr(), dh_o.data_ptr()
    for k_ in ('d_gi_h', 'd_gi_o', 'd_gh_h', 'd_gh_o', 'd_u_h', 'd_u_o', 'd_pre_h', 'd_pre_o'):
        setattr(b, k_, out[k_].data_ptr())
    n_scr = int(K.lib.twog_segrnn_bwd_persistent_scratch_bytes(C.byref(s)))
    scr = torch.empty(n_scr // 4 + 64, dtype=torch.float32, device=DEV)
    for rep in range(2):
        sync = torch.zeros(n_sync, dtype=torch.int32, device=DEV)
        rc = K.lib.twog_segrnn_bwd_persistent(C.byref(s), C.byref(b), scr.data_ptr(), scr.numel() * 4, sync.data_ptr(), K._stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
    st = sync[2048:2048 + 64].cpu().view(4, 16).float() * 0.01 / T
    Q1 = {0: 'wait X1', 1: 'loads+products', 2: 'combine', 3: 'd_pre', 4: 'dL/dw shares', 5: 'drain+signal'}
    Q2 = {0: 'wait X2', 1: 'sender-MLP product + dw sums', 2: 'softmax bwd / carry', 3: 'gate bwd, stores', 4: 'drain+signal',
          5: 'wait own X1', 6: 'W_hh product'}
    print(f'bs={bs} T={T} H={H} O={O} h={h}: BACKWARD us per step by role and phase')
    for r, name in enumerate(('Q1h', 'Q1o', 'Q2h', 'Q2o')):
        names = Q1 if r < 2 else Q2
        row = ', '.join(f'{names[k]} {st[r][k]:.2f}' for k in range(7) if k in names and st[r][k] > 0)
        print(f'  {name}: total {st[r][:7].sum():.2f} | {row}')
