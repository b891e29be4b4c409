#!/usr/bin/env python3
"""Few-tile, long-K GEMM launches of the recurrent chains in isolation: average launch time over a stream of launches
whose A operand rotates through fresh buffers (as in the chains, where every step's A was just written).
The floor printed beside each time is structural: a 32x32 accumulator block over K is one wave's serial MFMA chain
(K/2 instructions x 64 cycles), a 64x64 tile is four of them on the four SIMDs of ONE CU, so a launch with at most one
tile per CU cannot finish before 64*64*K*2 / 614 GFLOP/s however few tiles it has.
usage: [TWOG_GEMM_KS=0] [TWOG_GEMM_XSPLIT=n] python3 tools/gemm_chain_bench.py [chain]   (chain: twog_gemm_f32_chain, i.e.
with the reduction split over workgroups where the library's model -- or TWOG_GEMM_XSPLIT -- says so)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda'
CHAIN = len(sys.argv) > 1 and sys.argv[1] == 'chain'
SHAPES = [  # M, N, K, b_kmajor, note
    (1408, 512, 1536, True, 'BiGRU backward carry (176 tiles)'),
    (1280, 512, 1536, True, 'segment backward carry (160 tiles)'),
    (1280, 1024, 512, False, 'segment sender MLPs (320 tiles)'),
    (176, 512, 1536, True, '8 clips: backward carry (24 tiles)'),
    (176, 1536, 512, False, '8 clips: W_hh projection (72 tiles)'),
    (1280, 1024, 1536, True, 'segment d_mg = d_gi W_ihm (320 tiles)'),
    (96, 512, 1536, True, '8 clips: segment backward carry (24 tiles)'),
    (96, 1024, 1536, True, '8 clips: segment d_mg (48 tiles)'),
    (96, 1024, 512, False, '8 clips: sender MLPs (48 tiles)'),
    (96, 1536, 1024, False, '8 clips: W_ihm projection (72 tiles)'),
]
for M, N, Kk, bkm, note in SHAPES:
    As = [torch.randn(M, Kk, device=dev) for _ in range(16)]
    B = torch.randn((Kk, N) if bkm else (N, Kk), device=dev)
    C = torch.empty(M, N, device=dev)
    for i in range(20):
        K.gemm([dict(A=As[i % 16], B=B, C=C)], b_kmajor=bkm, split_k_workspace=False, chain=CHAIN)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(128):
            K.gemm([dict(A=As[i % 16], B=B, C=C)], b_kmajor=bkm, split_k_workspace=False, chain=CHAIN)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    ref = As[127 % 16].double() @ (B.double() if bkm else B.double().t())
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    floor = 64 * 64 * Kk * 2 / 614e3 * -(-(-(-M // 64) * -(-N // 64)) // 256)
    print(f'{note:44s} {M}x{N}x{Kk}: {e0.elapsed_time(e1) / 512 * 1e3:6.1f} us  (MFMA floor of the tile rounds {floor:5.1f} us)  class {K.gemm_last_class():#x} err {err:.1e}')
