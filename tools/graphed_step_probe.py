#!/usr/bin/env python3
"""Is the small-batch step host-bound? Times the bench step (a) eager, (b) eager with host enqueue vs device drain split,
(c) the whole step captured into ONE hipGraph (torch.cuda.CUDAGraph; the library's inner loop graphs off so that their
launches are captured directly).  usage: TWOG_NO_GRAPHS=1 python3 tools/graphed_step_probe.py [c2|c3|c5]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
wl = bench.select_workload(sys.argv[1] if len(sys.argv) > 1 else 'c2')
import twog_gcn_amd  # noqa
from twog_gcn_amd.models import TGGCN
from twog_gcn_amd.distributed import DataParallel, FusedAdam
from twog_gcn_amd.losses import select_loss
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = TGGCN(input_size=(2048 + 4 * bench.N_NODES, 2048), num_classes=(bench.N_CLASSES, None), **bench.CFG).to(dev).train()
dp = DataParallel(model)
opt = FusedAdam(dp.flat, lr=1e-4)
bs = bench.BS
x_human, x_objects, mask, targets = bench.synthetic_batch(bs, dev, 1234)
seg = torch.ones(bs, bench.T, bench.H, device=dev)
crit, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))
st = torch.zeros(bs, bench.T, bench.H, device=dev)
lt = [st, st, targets[0], targets[1], targets[0], targets[1]]
noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((bench.T * bench.O, bs, 2)).to(dev)
model._gumbel_noise_override = noise   # (the CPU draw + copy is not capturable; a device-side draw would replace it)

def step():
    dp.zero_grad()
    out = model(x_human, x_objects, mask, human_segmentation=seg)
    loss = sum(crit(out, lt))
    loss.backward()
    dp.all_reduce_gradients()
    opt.step(dp.grad_scale)
    return loss.detach()

for _ in range(8):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'eager: {(t2 - t0) / 10 * 1e3:.2f} ms/step (host enqueue {(t1 - t0) / 10 * 1e3:.2f} ms/step)')
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f'whole step as one graph: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step, loss {float(loss):.4f}')
