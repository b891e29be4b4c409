#!/usr/bin/env python3
"""dW = dY^T X launches of the 128x128 bf16x3 class with and without the fused column sums of dY (twog_gemm_t::a_colsum),
beside the separate column-sum launch they replace: ms per launch (20 launches between two events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda'
g = torch.Generator().manual_seed(0)
for M, N, Kk in ((1536, 512, 61440), (512, 2048, 61440), (1536, 2048, 76800), (2048, 4352, 7680), (512, 512, 61440)):
    A = torch.randn(Kk, M, generator=g).to(dev)
    B = (torch.randn(Kk, N, generator=g) * 0.1).to(dev)
    C = torch.empty(M, N, device=dev)
    cs = torch.empty(M, device=dev)
    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    plain = t(lambda: K.gemm([dict(A=A, B=B, C=C)], a_kmajor=True, b_kmajor=True))
    fused = t(lambda: K.gemm([dict(A=A, B=B, C=C, colsum=cs)], a_kmajor=True, b_kmajor=True))
    alone = t(lambda: K.colsum(A, out=cs))
    print(f'dW {M}x{N}x{Kk}: plain {plain:.3f} ms, with column sums {fused:.3f} ms (+{(fused / plain - 1) * 100:.1f} %), separate column-sum launch {alone:.3f} ms')
