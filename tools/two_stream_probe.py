#!/usr/bin/env python3
"""Probe (round 6): the 64-clip training step as TWO half-batch passes on two streams driven by two host threads, against one
64-clip pass. The recurrence chains (59 % of the step) are latency-bound and leave compute units idle; two independent chains
side by side should cost little more than one. No BatchNorm / gradient coupling here: this only prices the idea.
usage: python3 tools/two_stream_probe.py [clips=64] [steps=10]"""
import copy
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd.models import TGGCN  # noqa: E402
from twog_gcn_amd.losses import select_loss  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
torch.manual_seed(0)
T, H = bench.T, bench.H
criterion, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))


def make(nb, seed):
    m = TGGCN(input_size=(2048 + 4 * bench.N_NODES, 2048), num_classes=(bench.N_CLASSES, None), **bench.CFG).to(dev).train()
    x_human, x_objects, mask, targets = bench.synthetic_batch(nb, dev, seed=seed)
    seg = torch.ones(nb, T, H, device=dev)
    st = torch.zeros(nb, T, H, device=dev)
    lt = [st, st, targets[0], targets[1], targets[0], targets[1]]

    def step():
        for p in m.parameters():
            p.grad = None
        out = m(x_human, x_objects, mask, human_segmentation=seg)
        sum(criterion(out, lt)).backward()
    return step


def timed(fns, streams):
    def run(fn, st, n):
        with torch.cuda.stream(st):
            for _ in range(n):
                fn()
    for n in (3, steps):   # warm-up, then timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(f, s, n)) for f, s in zip(fns, streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / steps * 1e3


one = make(bs, 1)
ms1 = timed([one], [torch.cuda.Stream()])
print(f'one pass of {bs} clips:            {ms1:7.2f} ms per step = {bs / ms1 * 1e3:7.1f} clips/s', flush=True)
del one
torch.cuda.empty_cache()
for parts in (2, 4):
    fns = [make(bs // parts, 10 + i) for i in range(parts)]
    ms = timed(fns, [torch.cuda.Stream() for _ in range(parts)])
    print(f'{parts} passes of {bs // parts} clips side by side: {ms:7.2f} ms per step = {bs / ms * 1e3:7.1f} clips/s', flush=True)
    half = timed(fns[:1], [torch.cuda.Stream()])
    print(f'   (one pass of {bs // parts} clips alone:  {half:7.2f} ms)', flush=True)
    del fns
    torch.cuda.empty_cache()
