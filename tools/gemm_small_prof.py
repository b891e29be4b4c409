#!/usr/bin/env python3
"""Kernel-only timing model of the 64x64-tile GEMM class (recurrent steps): sweeps tiles per launch and K.
Run under rocprofv3 --kernel-trace (tools/gemm_small_prof.sh); prints the launch list that the shell script joins with
the trace (dispatch order = list order, REPS launches per configuration)."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels

K = get_kernels()
dev = 'cuda:0'
REPS = 10
configs = []
for groups, M, N in ((1, 512, 2048), (2, 512, 2048), (4, 512, 2048)):
    for Kk in (32, 64, 128, 256, 512, 1024):
        configs.append((groups, M, N, Kk, False))
configs.append((4, 512, 512, 1536, True))
out = []
for groups, M, N, Kk, bkm in configs:
    A = [torch.randn(M, Kk, device=dev) for _ in range(groups)]
    B = [torch.randn((Kk, N) if bkm else (N, Kk), device=dev) for _ in range(groups)]
    C = [torch.empty(M, N, device=dev) for _ in range(groups)]
    for _ in range(REPS):
        K.gemm([dict(A=a, B=b, C=c) for a, b, c in zip(A, B, C)], a_kmajor=False, b_kmajor=bkm)
    torch.cuda.synchronize()
    out.append(dict(groups=groups, M=M, N=N, K=Kk, bkm=bkm, tiles=groups * ((M + 63) // 64) * ((N + 63) // 64)))
json.dump(dict(reps=REPS, configs=out), open('gpurun_out/gemm_small_cfg.json', 'w'))
