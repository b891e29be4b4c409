#!/usr/bin/env python3
"""Bisect of a full-size gradient deviation: the SAME host composition (ops.py) runs once on the HIP kernels and once on
the torch test double (tests/fake_kernels.py, CPU); the operands of the frame-level attention backward are captured on
both sides and compared tensor by tensor.   usage: python3 tools/bisect_full_size.py [bs T H O N h seed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: F401,E402
from twog_gcn_amd import kernels as twog_kernels  # noqa: E402
from twog_gcn_amd.models import TGGCN  # noqa: E402
from tests.fake_kernels import FakeKernels  # noqa: E402
from tests.test_parity_gpu import STAGE1, _synthetic  # noqa: E402

a = [int(v) for v in sys.argv[1:8]] + [2, 120, 2, 8, 34, 512, 7][len(sys.argv) - 1:]
bs, T, H, O, N, h, seed = a
nF = bs * T


def run(dev, backend):
    twog_kernels._set_backend_for_tests(backend)
    K = twog_kernels.get_kernels()
    cap = {}
    orig_b, orig_g = K.attn_bwd, K.gemm

    def attn_bwd(descs):
        orig_b(descs)
        d = descs[0]
        if d['f']['n_inst'] == nF:
            for k, v in list(d.items()) + [('f.' + k2, v2) for k2, v2 in d['f'].items()]:
                if torch.is_tensor(v):
                    cap[k] = v.detach().cpu().clone()
    K.attn_bwd = attn_bwd
    torch.manual_seed(seed)
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(13, None), hidden_size=h, gcn_node=N, **STAGE1)
    x_human, x_objects, mask = _synthetic(bs, T, H, O, N, seed)
    seg = torch.ones(bs, T, H)
    noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, bs, 2))
    m = m.to(dev).train()
    m._gumbel_noise_override = noise
    out = m(x_human.to(dev), x_objects.to(dev), mask.to(dev), human_segmentation=seg.to(dev))
    rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)) for i, o in enumerate(out)]
    sum((o * r.to(dev)).sum() for o, r in zip(out, rs) if o.requires_grad).backward()
    K.attn_bwd = orig_b
    grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    return cap, grads


cap_g, gr_g = run('cuda:0', None)
cap_c, gr_c = run('cpu', FakeKernels())
print('frame-level attention backward operands: relative max deviation HIP vs test double')
for k in sorted(cap_c):
    c, g = cap_c[k], cap_g[k]
    scale = max(float(c.abs().max()), 1e-12)
    print(f'  {k:12s} {float((c - g).abs().max()) / scale:10.2e}   scale {scale:.3e}')
print('parameter gradients:')
rows = sorted(((float((gr_c[n] - gr_g[n]).abs().max()) / max(float(gr_c[n].abs().max()), 1e-12), n) for n in gr_c), reverse=True)
for e, n in rows[:10]:
    print(f'  {e:10.2e}  {n}')
c, g = cap_c['dmsg_oo'], cap_g['dmsg_oo']
diff = (c - g).abs()
rows_bad = (diff.max(dim=1).values > 1e-6 * float(c.abs().max())).nonzero().flatten()
print('dmsg_oo rows off:', len(rows_bad), 'of', c.shape[0], '-> (inst, sender):',
      [(int(r) // O, int(r) % O) for r in rows_bad[:24]])
if len(rows_bad):
    r = int(rows_bad[0])
    cols = (diff[r] > 1e-6 * float(c.abs().max())).nonzero().flatten()
    print('first bad row', r, 'bad cols', len(cols), cols[:16].tolist())
    print('  fake', c[r, cols[:8]].tolist())
    print('  hip ', g[r, cols[:8]].tolist())
    print('  msg ', cap_c['f.msg_oo'][r, cols[:8]].tolist())
    inst = r // O
    natt = H * H + 2 * H * O + O * O
    w = cap_c['f.att'][inst, H * H + 2 * H * O:].view(O, O)
    print('  att column of this sender', w[:, r % O].tolist())
    print('  mask', cap_c['f.obj_mask'].tolist())
