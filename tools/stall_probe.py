#!/usr/bin/env python3
"""Why do some steps take ~90 ms longer? Logs per-step wall time with allocator counters (segments allocated from the
driver) and Python GC runs. usage: python tools/stall_probe.py [n_steps]"""
import gc
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import twog_gcn_amd  # noqa
from twog_gcn_amd.models import TGGCN
from twog_gcn_amd.distributed import DataParallel, FusedAdam
from twog_gcn_amd.losses import select_loss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = TGGCN(input_size=(2048 + 4 * B.N_NODES, 2048), num_classes=(B.N_CLASSES, None), **B.CFG).to(dev).train()
dp = DataParallel(model)
opt = FusedAdam(dp.flat, lr=1e-4)
xh, xo, mask, tg = B.synthetic_batch(B.BS, dev, seed=1234)
seg = torch.ones(B.BS, B.T, B.H, device=dev)
crit, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))
st = torch.zeros(B.BS, B.T, B.H, device=dev)
tgts = [st, st, tg[0], tg[1], tg[0], tg[1]]
gc_runs = [0]
gc.callbacks.append(lambda phase, info: gc_runs.__setitem__(0, gc_runs[0] + (phase == 'start' and info['generation'] == 2)))
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dp.zero_grad()
    out = model(xh, xo, mask, human_segmentation=seg)
    sum(crit(out, tgts)).backward()
    opt.step(dp.grad_scale)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = torch.cuda.memory_stats()
    print(f'step {i:3d} {dt * 1e3:7.1f} ms  segments {ms["segment.all.allocated"]:5d} dev_alloc {ms.get("num_device_alloc", -1)} '
          f'dev_free {ms.get("num_device_free", -1)} retries {ms["num_alloc_retries"]} reserved {ms["reserved_bytes.all.current"] / 2**30:6.1f} GiB '
          f'gen2 gc {gc_runs[0]}', flush=True)
