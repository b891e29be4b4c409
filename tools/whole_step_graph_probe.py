#!/usr/bin/env python3
"""Can a whole training step (forward + criterion + backward + fused Adam) of a HOST-COMPOSED configuration be captured
into one torch.cuda.CUDAGraph (= hipGraph) and replayed? For the shipped configuration the step is GPU-bound and direct
launches win (profiles/HISTORY.md section 5); the constructor-default configurations are composed launch by launch from Python and are
host-bound. Prints eager vs replay time per step and checks that N replayed steps leave the same parameters as N eager ones.

usage: python tools/whole_step_graph_probe.py [defaults_seg|defaults|c2] [steps]"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd.models import TGGCN  # noqa: E402
from twog_gcn_amd.distributed import DataParallel, FusedAdam  # noqa: E402
from twog_gcn_amd.losses import select_loss  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'defaults_seg'
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
STAGE1 = dict(attention_style='v3', discrete_optimization_strategy='gs', filter_discrete_updates=False,
              message_humans_to_human=True, message_human_to_objects=True, message_objects_to_human=True,
              message_objects_to_object=True, message_geometry_to_objects=True, message_geometry_to_human=False,
              message_segment=True, message_type='v2', message_granularity='v1', message_aggregation='att',
              object_segment_update_strategy='ind', update_segment_threshold=0.5, bias=True, cat_level_states=0,
              share_level_mlps=0, add_segment_length=0, add_time_position=0, time_position_strategy='s',
              positional_encoding_style='e', discrete_networks_num_layers=1)
CFGS = {'defaults': dict(h=128, cfg={}), 'defaults_seg': dict(h=128, cfg=dict(message_segment=True)), 'c2': dict(h=512, cfg=STAGE1)}
w = CFGS[name]
bs, T, H, O, N, C = 8, 120, 2, 4, 26, 13
dev = torch.device('cuda', 0)


def build():
    torch.manual_seed(0)
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(C, None), hidden_size=w['h'], gcn_node=N, **w['cfg']).to(dev).train()
    dp = DataParallel(m)
    return m, dp, FusedAdam(dp.flat, lr=1e-4)


g = torch.Generator().manual_seed(1)
xh = torch.cat([torch.relu(torch.randn(bs, T, H, 2048, generator=g)), torch.rand(bs, T, 1, 4 * N, generator=g).expand(bs, T, H, 4 * N)], -1).contiguous().to(dev)
xo = torch.relu(torch.randn(bs, T, O, 2048, generator=g)).to(dev)
mask = torch.ones(bs, O, device=dev)
seg = torch.ones(bs, T, H, device=dev)
tg = [torch.randint(0, C, (bs, T, H), generator=g).to(dev) for _ in range(2)]
zt = torch.zeros(bs, T, H, device=dev)
targets = [zt, zt, tg[0], tg[1], tg[0], tg[1]]
criterion, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))
# the same Gumbel noise in every step on both sides (a captured graph would replay the noise it drew at capture time)
noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, bs, 2)).to(dev)


def make(m, dp, opt):
    m._gumbel_noise_override = noise

    def step():
        dp.zero_grad()
        out = m(xh, xo, mask, human_segmentation=seg)
        loss = sum(criterion(out, targets))
        loss.backward()
        opt.step(dp.grad_scale)
        return loss.detach()
    return step


m1, dp1, opt1 = build()
eager = make(m1, dp1, opt1)
for _ in range(3):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    eager()
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / n_steps

m2, dp2, opt2 = build()
gstep = make(m2, dp2, opt2)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):   # warm-up on the capture stream (allocator pools, library workspaces)
    for _ in range(3):
        gstep()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph, stream=side):
        gstep()
except Exception as e:  # noqa: BLE001
    print('capture failed:', type(e).__name__, str(e)[:400])
    sys.exit(1)
torch.cuda.synchronize()
# m2 has taken 3 eager steps + the capture pass does not execute; replay n_steps; m1 took 3 + n_steps eager
t0 = time.perf_counter()
for _ in range(n_steps):
    graph.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / n_steps
worst = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(m1.parameters(), m2.parameters()))
print(f'{name}: eager {t_eager * 1e3:.1f} ms/step, whole-step graph replay {t_graph * 1e3:.1f} ms/step; '
      f'parameters after {3 + n_steps} steps differ by at most {worst:.3e}')
