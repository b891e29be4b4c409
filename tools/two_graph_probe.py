#!/usr/bin/env python3
"""Probe (round 6): the 64-clip training step as P half- / quarter-batch passes CAPTURED into hipGraphs (forward + criterion +
backward each) and replayed side by side on P streams -- no host work between launches, so what is measured is whether the chip
overlaps two independent latency-bound recurrence chains. (Two host threads are GIL-bound: 91.6 ms for 2 x 32 clips against
64.7 ms for 64; two processes time-slice: 958 against 983 clips/s.) No BatchNorm / gradient coupling: this only prices the idea.
usage: python3 tools/two_graph_probe.py [clips=64] [replays=10]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd.models import TGGCN  # noqa: E402
from twog_gcn_amd.losses import select_loss  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
T, H, O = bench.T, bench.H, bench.O
criterion, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))


def make(nb, seed):
    torch.manual_seed(0)
    m = TGGCN(input_size=(2048 + 4 * bench.N_NODES, 2048), num_classes=(bench.N_CLASSES, None), **bench.CFG).to(dev).train()
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    x_human, x_objects, mask, targets = bench.synthetic_batch(nb, dev, seed=seed)
    seg = torch.ones(nb, T, H, device=dev)
    st = torch.zeros(nb, T, H, device=dev)
    lt = [st, st, targets[0], targets[1], targets[0], targets[1]]
    m._gumbel_noise_override = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, nb, 2)).to(dev)

    def step():
        out = m(x_human, x_objects, mask, human_segmentation=seg)
        sum(criterion(out, lt)).backward()
    return step


def capture(fn, stream):
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    torch.cuda.synchronize()
    return g


def replay(graphs, streams):
    for n in (2, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for g, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / reps * 1e3


for parts in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    fns = [make(bs // parts, 10 + i) for i in range(parts)]
    try:
        graphs = [capture(f, s) for f, s in zip(fns, streams)]
    except Exception as e:  # noqa: BLE001
        print(f'{parts} parts: capture failed: {type(e).__name__}: {str(e)[:300]}', flush=True)
        continue
    ms = replay(graphs, streams)
    alone = replay(graphs[:1], streams[:1])
    print(f'{parts} graph(s) of {bs // parts} clips side by side: {ms:7.2f} ms per {bs} clips = {bs / ms * 1e3:7.1f} clips/s   '
          f'(one graph of {bs // parts} clips alone: {alone:7.2f} ms)', flush=True)
    del graphs, fns
    torch.cuda.empty_cache()
