#!/usr/bin/env python3
"""Host-side time spent inside each kernel-interface call during one step (no device sync inside the step): shows
whether the launch loops run ahead of the GPU."""
import collections
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
if len(sys.argv) > 1:
    B.select_workload(sys.argv[1])   # c2 / c5 / ...: BASELINE configurations other than the bs64 headline
import twog_gcn_amd  # noqa
from twog_gcn_amd.models import TGGCN
from twog_gcn_amd.kernels import get_kernels
from twog_gcn_amd.distributed import DataParallel, FusedAdam
from twog_gcn_amd.losses import select_loss

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
K = get_kernels()
acc = collections.defaultdict(lambda: [0.0, 0])
for name in dir(K):
    fn = getattr(K, name)
    if name.startswith('_') or not callable(fn) or name in ('empty', 'zeros', 'workspace', 'version'):
        continue
    def wrap(fn=fn, name=name):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            acc[name][0] += time.perf_counter() - t
            acc[name][1] += 1
            return r
        return w
    setattr(K, name, wrap())


def step():
    dp.zero_grad()
    t0 = time.perf_counter()
    out = model(xh, xo, mask, human_segmentation=seg)
    t1 = time.perf_counter()
    loss = sum(crit(out, tgts))
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step(dp.grad_scale)
    return t1 - t0, t2 - t1, t3 - t2


for _ in range(8):   # (past the persistent launches' first, immediately checked calls)
    step()
torch.cuda.synchronize()
acc.clear()
t = time.perf_counter()
f, l, b = step()
host = time.perf_counter() - t
torch.cuda.synchronize()
total = time.perf_counter() - t
print(f'host: forward {f * 1e3:.1f} ms, loss {l * 1e3:.1f} ms, backward {b * 1e3:.1f} ms; host total {host * 1e3:.1f} ms, step incl. GPU drain {total * 1e3:.1f} ms')
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'  {k:26s} {s * 1e3:8.2f} ms host  {n:4d} calls  {s / n * 1e6:8.1f} us/call')
