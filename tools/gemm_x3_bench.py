#!/usr/bin/env python3
"""128x128-class GEMM launches in the four operand layouts: error against an fp64 product and TFLOP/s (fp32-equivalent:
2 M N K per launch).  usage: [TWOG_GEMM_X3=1] python3 tools/gemm_x3_bench.py     (run once with and once without)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda'
# (M, N, K, a_kmajor, b_kmajor, note)
SHAPES = [(61440, 1536, 512, False, False, 'forward NN (GRU input projection)'),
          (61440, 512, 2048, False, False, 'forward NN (embedding)'),
          (61440, 512, 1536, False, True, 'dX NT'),
          (7680, 4352, 2048, False, True, 'dX NT (geometry MLP)'),
          (1536, 512, 61440, True, True, 'dW TT (split-K)'),
          (512, 2048, 61440, True, True, 'dW TT (split-K)'),
          (2048, 4352, 7680, True, True, 'dW TT (geometry MLP)'),
          (4096, 4096, 4096, True, False, 'TN'),
          (300, 200, 96, False, False, 'ragged edges (forced 128 class by TWOG_GEMM_TILE=128 only)')]
g = torch.Generator().manual_seed(0)
for M, N, Kk, akm, bkm, note in SHAPES:
    A = torch.randn((Kk, M) if akm else (M, Kk), generator=g).to(dev)
    B = (torch.randn((Kk, N) if bkm else (N, Kk), generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        K.gemm([dict(A=A, B=B, C=C, bias=bias, act=0)], a_kmajor=akm, b_kmajor=bkm)
    cls = K.gemm_last_class()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.gemm([dict(A=A, B=B, C=C, bias=bias, act=0)], a_kmajor=akm, b_kmajor=bkm)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    # fp64 reference on a slab of rows (the whole product for the small ones)
    rows = min(M, 2048)
    Ad = (A[:, :rows].t() if akm else A[:rows]).double()
    Bd = (B if bkm else B.t()).double()
    ref = Ad @ Bd + bias.double()
    err = float((C[:rows].double() - ref).abs().max() / ref.abs().max())
    print(f'{note:46s} {M}x{N}x{Kk} {"T" if akm else "N"}{"T" if bkm else "N"}: {ms:7.3f} ms {2.0 * M * N * Kk / ms / 1e9:7.1f} TF  class {cls:#x}  max err / max|ref| {err:.2e}')
