#!/usr/bin/env python3
"""One GEMM shape in a loop, for rocprofv3 --pmc runs.  python tools/gemm_pmc.py M N K [akm bkm [nows]]
(nows: no split-K workspace, the form the recurrent chains use -> k-split / 32-row kernels)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
M, N, Kk = (int(a) for a in sys.argv[1:4])
akm, bkm = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (0, 0)
ws = not (len(sys.argv) > 6 and sys.argv[6] == 'nows')
A = torch.randn((Kk, M) if akm else (M, Kk), device='cuda')
B = torch.randn((Kk, N) if bkm else (N, Kk), device='cuda')
C = torch.empty(M, N, device='cuda')
for _ in range(5):
    K.gemm([dict(A=A, B=B, C=C)], a_kmajor=bool(akm), b_kmajor=bool(bkm), split_k_workspace=ws)
torch.cuda.synchronize()
