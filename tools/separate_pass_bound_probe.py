#!/usr/bin/env python3
"""Probe (round 6): what is the MOST the step could gain from fusing each family of separate passes into a
neighbouring launch? Each family is switched OFF in turn (its launches simply do not happen: the results of such a step are
wrong, only its duration means anything) and the 64-clip training step is timed against the unmodified one on the same box.
A family whose removal does not move the step is hidden behind something else (the side stream, a chain) and not worth a fusion.
usage: python3 tools/separate_pass_bound_probe.py [workload=c3] [steps=10]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd import kernels as twog_kernels  # noqa: E402
from twog_gcn_amd.models import TGGCN  # noqa: E402
from twog_gcn_amd.losses import select_loss  # noqa: E402

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
torch.manual_seed(0)
T, H = bench.T, bench.H
criterion, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))
m = TGGCN(input_size=(2048 + 4 * bench.N_NODES, 2048), num_classes=(bench.N_CLASSES, None), **bench.CFG).to(dev).train()
x_human, x_objects, mask, targets = bench.synthetic_batch(bs, dev, seed=1)
seg = torch.ones(bs, T, H, device=dev)
st = torch.zeros(bs, T, H, device=dev)
lt = [st, st, targets[0], targets[1], targets[0], targets[1]]


def step():
    for p in m.parameters():
        p.grad = None
    out = m(x_human, x_objects, mask, human_segmentation=seg)
    sum(criterion(out, lt)).backward()


def timed():
    for n in (3, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / steps * 1e3


HK = twog_kernels.HipKernels
orig = {k: getattr(HK, k) for k in ('relu_bwd', 'rank1_update', 'colsum', 'colsum_many', 'fill_zero', 'rowops')}


def no_relu_bwd(self, dy, y, dx=None):
    return dy if dx is None else dx


def no_rank1(self, dst, s, v):
    return None


def no_colsum(self, x, rowscale=None, out=None, accumulate=False):
    return out if out is not None else torch.empty(x.shape[-1], dtype=torch.float32, device=x.device)


def no_colsum_many(self, ops):
    return None


def no_fill_zero(self, t):
    return t


def rowops_without(kinds):
    def f(self, ops):
        return orig['rowops'](self, [o for o in ops if o[0] not in kinds])
    return f


FAMILIES = [
    ('relu_bwd (separate launches)', dict(relu_bwd=no_relu_bwd, rowops=rowops_without({'relu_bwd'}))),
    ('rank1 updates', dict(rank1_update=no_rank1, rowops=rowops_without({'rank1'}))),
    ('column sums outside the dW launches (colsum, colsum_many)', dict(colsum=no_colsum, colsum_many=no_colsum_many)),
    ('fill_zero', dict(fill_zero=no_fill_zero)),
    ('all of the above', dict(relu_bwd=no_relu_bwd, rank1_update=no_rank1, colsum=no_colsum, colsum_many=no_colsum_many,
                              fill_zero=no_fill_zero, rowops=rowops_without({'relu_bwd', 'rank1'}))),
]

base = timed()
print(f'unmodified step ({bs} clips, forward + loss + backward, no optimiser): {base:7.2f} ms', flush=True)
for name, patch in FAMILIES:
    for k, v in patch.items():
        setattr(HK, k, v)
    try:
        ms = timed()
    finally:
        for k, v in orig.items():
            setattr(HK, k, v)
    again = timed()
    print(f'without {name:58s}: {ms:7.2f} ms  ({ms - (base + again) / 2:+.2f} ms against the unmodified step before / after: '
          f'{base:.2f} / {again:.2f})', flush=True)
    base = again
