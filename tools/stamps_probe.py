"""GPU box, diagnostic library only (tools/stamps_probe.sh): cycles per k-tile and phase of the X3 chain kernels."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda'
PH = ['issue the loads of tile t+RS', 'fragment reads + MFMAs', 'vmcnt wait + split + ds_write', 'barrier', 'fragment reads alone (inside phase 1)']
for ku in (os.environ.get('TWOG_X3S_KU', '2'),):
    for M, N, Kk, bkm, note in ((1408, 512, 1536, True, 'BiGRU backward carry, 176 tiles, 8 waves (k-split)'),
                                (1920, 1024, 1536, True, 'segment d_mg + carry, 480 tiles, 4 waves'),
                                (1280, 1024, 512, False, 'segment sender MLPs, 320 tiles, 8 waves')):
        A = torch.randn(M, Kk, device=dev)
        B = torch.randn((Kk, N) if bkm else (N, Kk), device=dev)
        Cm = torch.empty(M, N, device=dev)
        for _ in range(5):
            K.gemm([dict(A=A, B=B, C=Cm)], b_kmajor=bkm, split_k_workspace=False, chain=False)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 128)()
        assert K.lib.twog_debug_stamps(buf) == 0
        cls = K.gemm_last_class()
        waves = 8 if cls & K.GEMM_KSPLIT else 4
        xk = 32 if cls & K.GEMM_KSPLIT else 16
        print(f'{note}: {M}x{N}x{Kk} class {cls:#x} (TWOG_X3S_KU={ku})')
        for w in range(waves):
            v = [buf[w * 8 + i] for i in range(5)]
            nkt = Kk // xk
            names = ['issue loads', 'frag reads', 'MFMAs', 'wait+split+ds_write', 'barrier']
            vals = [v[0], v[4], v[1], v[2], v[3]]
            print(f'  wave {w}: ' + '  '.join(f'{n} {x / nkt:6.0f}' for n, x in zip(names, vals)) + f'  | cycles per k-tile of {xk}: {sum(vals) / nkt:6.0f}')
