#!/usr/bin/env python3
"""Replays ONE parity_fuzz case on the GPU with every kernel call followed by a synchronize and a comparison of the
caller's input tensors against their host copies: names the first kernel call after which an input changed (an
out-of-bounds or aliased write), then prints the per-output deviation from the oracle.
usage: python3 tools/fuzz_case_probe.py <seed> <case index>"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import parity_fuzz as pf  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from twog_gcn_amd import kernels  # noqa: E402
from twog_gcn_amd.models import TGGCN  # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for i in range(target):
    pf.one_case(rng, i, dry=True)
cap = {}
orig = cpu_ref.tggcn_forward


def spy(sd, cfg, x_human, x_objects, mask, **kw):
    cap.update(sd=sd, cfg=cfg, x_human=x_human, x_objects=x_objects, mask=mask, kw=kw)
    return orig(sd, cfg, x_human, x_objects, mask, **kw)


cpu_ref.tggcn_forward = spy
os.environ['TWOG_FUZZ_ORACLE_ONLY'] = '1'
desc = pf.one_case(rng, target)
cpu_ref.tggcn_forward = orig
print(desc)
dev = 'cuda:0'
cfg = {k: v for k, v in cap['cfg'].items() if k not in ('input_size', 'num_classes', 'num_subactivities',
                                                         'num_affordances', 'object_input_size')}
m = TGGCN(input_size=(2048 + 4 * desc['N'], 2048), num_classes=desc['classes'], **cfg)
m.load_state_dict({k: v.detach() for k, v in cap['sd'].items()})
m = m.to(dev)
m.train(cap['kw']['training'])
m._gumbel_noise_override = cap['kw']['gumbel_noise']
kw_host = {k: v for k, v in cap['kw'].items() if k not in ('training', 'gumbel_noise', 'aux')}
kw_dev = {k: v.to(dev) for k, v in kw_host.items()}
watch = {k: (kw_dev[k], kw_host[k].clone()) for k in kw_dev}
xh, xo, mk = cap['x_human'].to(dev), cap['x_objects'].to(dev), cap['mask'].to(dev)
watch.update(x_human=(xh, cap['x_human'].clone()), x_objects=(xo, cap['x_objects'].clone()), mask=(mk, cap['mask'].clone()))
real = kernels.get_kernels()
reported = set()


class Probe:
    name = 'hip'

    def __getattr__(self, attr):
        f = getattr(real, attr)
        if not callable(f):
            return f

        def call(*a, **k):
            r = f(*a, **k)
            torch.cuda.synchronize()
            for n, (d, hcopy) in watch.items():
                if n not in reported and not torch.equal(d.cpu().float(), hcopy.float()):
                    reported.add(n)
                    shapes = [tuple(x.shape) for x in a if torch.is_tensor(x)]
                    print(f'INPUT {n} CHANGED after kernel call {attr} tensor-args {shapes}', flush=True)
                    if a and isinstance(a[0], dict):
                        print('   descriptor:', {kk: (tuple(vv.shape), vv.data_ptr()) if torch.is_tensor(vv) else vv
                                                 for kk, vv in a[0].items()})
                    print('   now:', d.flatten()[:16].tolist(), 'was:', hcopy.flatten()[:16].tolist())
            return r
        return call


kernels._set_backend_for_tests(Probe())
out = m(xh, xo, mk, **kw_dev)
ref = orig({k: v.detach() for k, v in cap['sd'].items()}, cap['cfg'], cap['x_human'], cap['x_objects'], cap['mask'],
           **cap['kw'])
for i, (a, b) in enumerate(zip(out, ref)):
    print('output', i, tuple(a.shape), (a.detach().cpu() - b.detach()).abs().max().item())
print('out0', out[0].flatten().tolist(), 'ref0', ref[0].flatten().tolist())
