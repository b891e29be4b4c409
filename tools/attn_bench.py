#!/usr/bin/env python3
"""GPU microbenchmark of the segment-level attention call (64 clips x 2 directions, h=512, H=2, O=8)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
dev = 'cuda:0'
bs, H, O, h = 64, 2, 8, 512


def desc():
    t = lambda *s: torch.randn(*s, device=dev)
    return dict(feat_h=t(bs * H, h), feat_o=t(bs * O, h), msg_hh=t(bs * H, 2 * h)[:, :h], msg_ho=t(bs * H, 2 * h)[:, h:],
                msg_oh=t(bs * O, 2 * h)[:, :h], msg_oo=t(bs * O, 2 * h)[:, h:], out_hh=t(bs * H, 2 * h)[:, :h],
                out_oh=t(bs * H, 2 * h)[:, h:], out_ho=t(bs * O, 2 * h)[:, :h], out_oo=t(bs * O, 2 * h)[:, h:],
                obj_mask=torch.ones(bs, O, device=dev), att=t(bs, H * H + 2 * H * O + O * O), n_inst=bs, inst_per_clip=1,
                H=H, O=O, D=h, hidden=h, scale=1 / math.sqrt(h), recv_mask_ho=0)


d = [desc(), desc()]
for _ in range(3):
    K.attn_fwd(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    K.attn_fwd(d)
e1.record()
torch.cuda.synchronize()
print('TWOG_ATTN_DBG', os.environ.get('TWOG_ATTN_DBG'), 'attn_fwd seg-level: %.1f us/call' % (e0.elapsed_time(e1) / 200 * 1e3))
