#!/usr/bin/env python3
"""GPU microbenchmark of twog_gemm_f32 on the GEMM shapes of the C3 workload (bs64, T=120), with torch.mm (rocBLAS)
as a yardstick. Usage (GPU box): python tools/gemm_bench.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels

K = get_kernels()
dev = 'cuda:0'


def timeit(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def run(name, M, N, Kk, akm, bkm, groups=1):
    A = torch.randn((Kk, M) if akm else (M, Kk), device=dev)
    B = torch.randn((Kk, N) if bkm else (N, Kk), device=dev)
    Cs = [torch.empty(M, N, device=dev) for _ in range(groups)]
    t = timeit(lambda: K.gemm([dict(A=A, B=B, C=c) for c in Cs], a_kmajor=akm, b_kmajor=bkm))
    Am = A.t() if akm else A
    Bm = B if bkm else B.t()
    t_ref = timeit(lambda: [torch.mm(Am, Bm) for _ in range(groups)])
    fl = 2.0 * M * N * Kk * groups
    print(f'{name:34s} M={M:6d} N={N:5d} K={Kk:6d} {"T" if akm else "N"}{"T" if bkm else "N"} x{groups}: '
          f'{t * 1e3:8.3f} ms {fl / t / 1e12:6.1f} TF | rocBLAS {t_ref * 1e3:8.3f} ms {fl / t_ref / 1e12:6.1f} TF', flush=True)


if __name__ == '__main__':
    print('TILE', os.environ.get('TWOG_GEMM_TILE'), 'SPLITK', os.environ.get('TWOG_GEMM_SPLITK'))
    run('object embedding fwd', 61440, 512, 2048, False, False)
    run('geometry mlp.0 fwd', 7680, 2048, 4352, False, False)
    run('object bigru gi fwd', 61440, 1536, 512, False, False)
    run('object seg gi fwd', 61440, 1536, 2048, False, False)
    run('object msg mlp fwd', 61440, 512, 1024, False, False)
    run('object seg gi dX', 61440, 2048, 1536, False, True)
    run('object embedding dW', 512, 2048, 61440, True, True)
    run('object seg W_ih dW', 1536, 2048, 61440, True, True)
    run('geometry mlp.0 dW', 2048, 4352, 7680, True, True)
    run('human msg mlp dW', 512, 1024, 15360, True, True)
    run('seg W_hh dW', 1536, 512, 60928, True, True)
    run('recurrent step objects gh', 512, 1536, 512, False, False, groups=4)
    run('recurrent step humans gh', 128, 1536, 512, False, False, groups=4)
    run('recurrent step bwd', 512, 512, 1536, False, True, groups=4)
