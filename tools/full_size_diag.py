#!/usr/bin/env python3
"""Per-parameter gradient error at a full-size shape: the HIP path AND the fp32 oracle, each against the oracle run in
fp64 -- tells a kernel-specific deviation from fp32 rounding that the CPU restatement shares.
usage: python3 tools/full_size_diag.py [bs T H O N h seed]      (default: 2 120 2 8 34 512 7 = BASELINE configs[2])"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: F401,E402
from twog_gcn_amd.models import TGGCN  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from tests.test_parity_gpu import STAGE1, _synthetic  # noqa: E402

a = [int(v) for v in sys.argv[1:8]] + [2, 120, 2, 8, 34, 512, 7][len(sys.argv) - 1:]
bs, T, H, O, N, h, seed = a
DEV = 'cuda:0'
torch.manual_seed(seed)
m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(13, None), hidden_size=h, gcn_node=N, **STAGE1)
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
x_human, x_objects, mask = _synthetic(bs, T, H, O, N, seed)
seg = torch.ones(bs, T, H)
noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, bs, 2))
rs = None


def oracle(dtype):
    global rs
    sd = {k: (v.detach().to(dtype).clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k
              else (v.detach().to(dtype).clone() if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
    out = cpu_ref.tggcn_forward(sd, dict(m.cfg), x_human.to(dtype), x_objects.to(dtype), mask.to(dtype),
                                human_segmentation=seg.to(dtype), training=True, gumbel_noise=noise.to(dtype))
    if rs is None:
        rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)) for i, o in enumerate(out)]
    sum((o * r.to(dtype)).sum() for o, r in zip(out, rs) if o.requires_grad).backward()
    return {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}, out


g32, o32 = oracle(torch.float32)
g64, o64 = oracle(torch.float64)
m = m.to(DEV).train()
m._gumbel_noise_override = noise
out = m(x_human.to(DEV), x_objects.to(DEV), mask.to(DEV), human_segmentation=seg.to(DEV))
sum((o * r.to(DEV)).sum() for o, r in zip(out, rs) if o.requires_grad).backward()
print('outputs: max |HIP - fp64|, |fp32 - fp64|:',
      [(round((o.detach().cpu().double() - r).abs().max().item(), 9), round((q.double() - r).abs().max().item(), 9))
       for o, q, r in zip(out, o32, o64)])
rows = []
for n, p in m.named_parameters():
    if n not in g64 or p.grad is None:
        continue
    ref = g64[n]
    scale = max(ref.abs().max().item(), 1e-12)
    rows.append(((p.grad.cpu().double() - ref).abs().max().item() / scale,
                 (g32[n].double() - ref).abs().max().item() / scale, scale, n))
rows.sort(reverse=True)
print('rel err vs fp64 oracle:   HIP        fp32 oracle   grad scale   parameter')
for e_hip, e_32, scale, n in rows:
    print(f'                      {e_hip:10.2e}  {e_32:10.2e}  {scale:10.2e}   {n}')
print(json.dumps(dict(shape=[bs, T, H, O, N, h], worst_hip=rows[0][0], worst_fp32=max(r[1] for r in rows))))
