#!/usr/bin/env python3
"""The four persistent launches (frame-level BiGRU forward / backward, segment-level forward / backward) of the library
TWOG_LIB_PATH points at -- the JITTER build (`make -C 2g-gcn_amd/csrc jitter`: a pseudo-random pause of 0 ... ~4 us per wave in
front of every publish and every poll) -- repeated `reps` times per shape; every word of every output of every repetition
must equal the reference file's (written by the same script run on the shipped library with `write`): the arithmetic is the
same, only the order in which workgroups reach their hand-offs changes from step to step and from launch to launch.
usage: python3 tools/persist_jitter_check.py write|check FILE REPS
(tests/test_kernels_gpu.py::test_persistent_hand_offs_hold_under_jitter runs both)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, path, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
os.environ['TWOG_PERSIST_CHECK'] = 'sync'
import torch  # noqa: E402
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd import kernels  # noqa: E402
from tests.test_kernels_gpu import _seg_params, rnd  # noqa: E402

K = kernels.get_kernels()
DEV = 'cuda:0'
# BASELINE configs[0] (one CAD-120 clip), configs[1] (8 clips, MPHOI layout, h = 512), configs[4] (16 clips, Bimanual layout,
# h = 64: eight chunks), T = 120 each
SHAPES = [(1, 120, 1, 5, 512), (8, 120, 2, 4, 512), (16, 120, 2, 9, 64)]
ref = torch.load(path) if mode == 'check' else {}
bad = 0
for bs, T, H, O, h in SHAPES:
    key = f'{bs}x{T}x{H}x{O}x{h}'
    pg = _seg_params(DEV, bs, T, H, O, h, (True, True, True, True), True)
    dh_h, dh_o = rnd(bs, T, H, 2 * h, seed=31).to(DEV), rnd(bs, T, O, 2 * h, seed=32).to(DEV)
    ws = 0.2 * math.sqrt(64.0 / h)
    types = [{k: v.to(DEV) for k, v in dict(gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=ws),
                                              b_hh_f=rnd(3 * h, seed=20 + i), w_hh_r=rnd(3 * h, h, seed=30 + i, scale=ws),
                                              b_hh_r=rnd(3 * h, seed=40 + i)).items()} for i, E in enumerate((H, O, 1))]
    for rep in range(reps if mode == 'check' else 1):
        out = {}
        b = K.segrnn_fwd(pg)
        assert K.last_segrnn_persistent, 'the persistent segment forward did not run'
        o = K.segrnn_bwd(pg, b, dh_h, dh_o)
        assert K.last_segrnn_bwd_persistent, 'the persistent segment backward did not run'
        for k in ['hs_h', 'hs_o', 'save_h', 'save_o', 'msrc_h', 'msrc_o', 'mg_h', 'mg_o', 'att']:
            out['seg_fwd.' + k] = b[k]
        for k in sorted(o):
            out['seg_bwd.' + k] = o[k]
        res = K.bigru_fwd(types, bs, T, h)
        assert K.last_bigru_persistent, 'the persistent BiGRU forward did not run'
        bt = [dict(d_out=rnd(*r[0].shape, seed=50 + i).to(DEV), save=r[1], out=r[0], w_hh_f=d['w_hh_f'], w_hh_r=d['w_hh_r'])
              for i, (r, d) in enumerate(zip(res, types))]
        g = K.bigru_bwd(bt, bs, T, h)
        assert K.last_bigru_bwd_persistent, 'the persistent BiGRU backward did not run'
        for i, ((oo, ss), (dgi, dgh)) in enumerate(zip(res, g)):
            out.update({f'bigru_fwd.out{i}': oo, f'bigru_fwd.save{i}': ss, f'bigru_bwd.d_gi{i}': dgi, f'bigru_bwd.d_gh{i}': dgh})
        torch.cuda.synchronize()
        if mode == 'write':
            ref[key] = {k: v.cpu() for k, v in out.items()}
        else:
            for k, v in out.items():
                if not torch.equal(v.cpu(), ref[key][k]):
                    bad += 1
                    n = int((v.cpu() != ref[key][k]).sum())
                    print(f'shape {key} repetition {rep}: {k} differs from the shipped library\'s result in {n} words', flush=True)
if mode == 'write':
    torch.save(ref, path)
    print('reference written:', {k: len(v) for k, v in ref.items()})
else:
    print(f'{reps} repetitions x {len(SHAPES)} shapes x 4 persistent launches under jitter: {bad} tensors differ')
    sys.exit(1 if bad else 0)
