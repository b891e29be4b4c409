#!/usr/bin/env python3
"""Diagnosis of one parity_fuzz case: per-parameter gradient error of the HIP path AND of the fp32 oracle against the
oracle run in fp64 (is a deviation rounding-level, i.e. shared by the fp32 CPU restatement, or specific to the kernels?).
usage: python3 tools/parity_fuzz_diag.py <seed> <case index>"""
import random
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from tools import parity_fuzz as pf  # noqa: E402
from oracle import cpu_ref  # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
captured = {}
orig = cpu_ref.tggcn_forward


def spy(sd, cfg, x_human, x_objects, mask, **kw):
    captured.update(sd=sd, cfg=cfg, x_human=x_human, x_objects=x_objects, mask=mask, kw=kw)
    return orig(sd, cfg, x_human, x_objects, mask, **kw)


cpu_ref.tggcn_forward = spy
import os  # noqa: E402
os.environ['TWOG_FUZZ_ORACLE_ONLY'] = '1'
for i in range(target + 1):
    d = pf.one_case(rng, i)
cpu_ref.tggcn_forward = orig
print(d)
c = captured


def run(dtype):
    sd = {k: (v.detach().to(dtype).requires_grad_(True) if v.is_floating_point() and 'running' not in k
              else (v.detach().to(dtype) if v.is_floating_point() else v.clone())) for k, v in c['sd'].items()}
    kw = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in c['kw'].items()}
    out = orig(sd, c['cfg'], c['x_human'].to(dtype), c['x_objects'].to(dtype), c['mask'].to(dtype), **kw)
    rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)).to(dtype) for i, o in enumerate(out)]
    sum((o * r).sum() for o, r in zip(out, rs) if o.requires_grad).backward()
    return {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}, out


g64, _ = run(torch.float64)
g32, _ = run(torch.float32)
from twog_gcn_amd.models import TGGCN  # noqa: E402
cfg = dict(c['cfg'])
N = cfg['gcn_node']
classes = (10, 12) if any('object_recognition' in k for k in c['sd']) else (13, None)
cfg.pop('num_subactivities', None)
cfg.pop('num_affordances', None)
m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=classes, **{k: v for k, v in cfg.items() if k not in ('object_input_size',)})
m.load_state_dict({k: v.detach() for k, v in c['sd'].items()})
m = m.to('cuda:0')
m.train(c['kw'].get('training', True))
m._gumbel_noise_override = c['kw']['gumbel_noise']
kw = {k: v.to('cuda:0') for k, v in c['kw'].items() if k in ('human_segmentation', 'objects_segmentation')}
out = m(c['x_human'].to('cuda:0'), c['x_objects'].to('cuda:0'), c['mask'].to('cuda:0'), **kw)
rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)) for i, o in enumerate(out)]
sum((o * r.to('cuda:0')).sum() for o, r in zip(out, rs) if o.requires_grad).backward()
rows = []
for n, p in m.named_parameters():
    if n not in g64 or p.grad is None:
        continue
    ref = g64[n]
    scale = max(ref.abs().max().item(), 1e-12)
    rows.append((((p.grad.cpu().double() - ref).abs().max().item()) / scale,
                 ((g32[n].double() - ref).abs().max().item()) / scale, scale, n))
rows.sort(reverse=True)
print('rel err vs fp64 oracle:   HIP        fp32 oracle   grad scale   parameter')
for e_hip, e_32, scale, n in rows[:12]:
    print(f'                      {e_hip:10.2e}  {e_32:10.2e}  {scale:10.2e}   {n}')
