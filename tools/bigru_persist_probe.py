#!/usr/bin/env python3
"""Frame-level BiGRU forward at the bench shape (64 clips: humans 2, objects 8, geometry 1; T = 120, h = 512) and at 8
clips (2 / 4 / 1): the launch-per-step path against the persistent launch (TWOG_BIGRU_PERSIST=1), ms per call.
usage: python3 tools/bigru_persist_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda'
T, h = 120, 512
SHAPES = ((64, (2, 8, 1)), (8, (2, 4, 1)), (16, (2, 9, 1)), (32, (2, 4, 1)), (24, (2, 8, 1)))
if os.environ.get('TWOG_PROBE_ONLY'):
    SHAPES = SHAPES[:1]
for bs, Es in SHAPES:
    g = torch.Generator().manual_seed(0)
    types = []
    for E in Es:
        types.append(dict(gi=torch.randn(bs, T, E, 6 * h, generator=g).to(dev),
                          w_hh_f=(torch.randn(3 * h, h, generator=g) * 0.07).to(dev), b_hh_f=torch.randn(3 * h, generator=g).to(dev),
                          w_hh_r=(torch.randn(3 * h, h, generator=g) * 0.07).to(dev), b_hh_r=torch.randn(3 * h, generator=g).to(dev)))
    res = {}
    for mode in ('0', '1', '0', '1'):
        os.environ['TWOG_BIGRU_PERSIST'] = mode
        for _ in range(3):
            out = K.bigru_fwd(types, bs, T, h)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = K.bigru_fwd(types, bs, T, h)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[mode] = out
        print(f'bs {bs} E {Es}: TWOG_BIGRU_PERSIST={mode}: {ms:7.3f} ms per call = {ms / T * 1e3:6.1f} us per time step')
    d = max(float((a[0] - b[0]).abs().max()) for a, b in zip(res['0'], res['1']))
    print(f'   max |persistent - stepwise| over the outputs: {d:.2e}')

# backward at 8 clips: launch-per-step (gate backward fused into the carry GEMM's epilogue) against the persistent launch
bs, Es = 8, (2, 4, 1)
g = torch.Generator().manual_seed(1)
types = []
for E in Es:
    types.append(dict(gi=torch.randn(bs, T, E, 6 * h, generator=g).to(dev),
                      w_hh_f=(torch.randn(3 * h, h, generator=g) * 0.07).to(dev), b_hh_f=torch.randn(3 * h, generator=g).to(dev),
                      w_hh_r=(torch.randn(3 * h, h, generator=g) * 0.07).to(dev), b_hh_r=torch.randn(3 * h, generator=g).to(dev)))
fw = K.bigru_fwd(types, bs, T, h)
bt = [dict(d_out=torch.randn(bs, T, E, 2 * h, generator=g).to(dev), save=sv, out=o, w_hh_f=y['w_hh_f'], w_hh_r=y['w_hh_r'])
      for (o, sv), y, E in zip(fw, types, Es)]
for mode in ('0', 'auto', '0', 'auto'):
    os.environ['TWOG_BIGRU_PERSIST'] = mode
    for _ in range(3):
        K.bigru_bwd(bt, bs, T, h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.bigru_bwd(bt, bs, T, h)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'backward bs {bs} E {Es}: TWOG_BIGRU_PERSIST={mode}: {ms:7.3f} ms per call = {ms / T * 1e3:6.1f} us per time step')
