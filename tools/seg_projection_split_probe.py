#!/usr/bin/env python3
"""Experiment: the segment level's per-step projection launch at 64 clips (both directions, humans 128 rows / objects 512
rows; gh = h_prev W_hh^T: K = 512, gim = m W_ihm^T: K = 1 024; 240 tiles of 128x128) as shipped, against the same launch
with the K = 1 024 products cut into two K = 512 halves through the batch field (separate outputs, 360 equal tiles).
usage: python3 tools/seg_projection_split_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda'
h, bs = 512, 64
rows = {'h': bs * 2, 'o': bs * 8}
NBUF = 8
hp = {k: [torch.randn(r, h, device=dev) for _ in range(NBUF)] for k, r in rows.items()}
mg = {k: [torch.randn(r, 2 * h, device=dev) for _ in range(NBUF)] for k, r in rows.items()}
whh = {(k, d): torch.randn(3 * h, h, device=dev) for k in rows for d in range(2)}
wih = {(k, d): torch.randn(3 * h, 5 * h, device=dev) for k in rows for d in range(2)}   # message columns [3h:5h]
bhh = {(k, d): torch.randn(3 * h, device=dev) for k in rows for d in range(2)}
gh = {(k, d): torch.empty(r, 3 * h, device=dev) for k, r in rows.items() for d in range(2)}
gim = {(k, d): torch.empty(2, r, 3 * h, device=dev) for k, r in rows.items() for d in range(2)}


def problems(i, split):
    ps = []
    for d in range(2):
        for k, r in rows.items():
            ps.append(dict(A=hp[k][(i + d) % NBUF], B=whh[(k, d)], C=gh[(k, d)], bias=bhh[(k, d)]))
            A, B, Cm = mg[k][(i + d) % NBUF], wih[(k, d)][:, 3 * h:], gim[(k, d)]
            if split:
                ps.append(dict(A=A[:, :h], B=B[:, :h], C=Cm[0], batch=(2, h, h, r * 3 * h)))
            else:
                ps.append(dict(A=A, B=B, C=Cm[0]))
    return ps


for split in (False, True, False, True):
    for i in range(10):
        K.gemm(problems(i, split), chain=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(64):
            K.gemm(problems(i, split), chain=True)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    k, d = 'o', 1
    ref = mg[k][(63 + d) % NBUF].double() @ wih[(k, d)][:, 3 * h:].double().t()
    got = gim[(k, d)][0].double() + (gim[(k, d)][1].double() if split else 0)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f'split={split}: {e0.elapsed_time(e1) / 512 * 1e3:6.1f} us per launch, class {K.gemm_last_class():#x}, err {err:.1e}')
