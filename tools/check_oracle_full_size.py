#!/usr/bin/env python3
"""Full-size pin of the oracle: the LIVE reference (imported from /root/reference) against oracle/cpu_ref.py at the
BASELINE width (h = 512, T = 120), outputs + every parameter gradient + side-by-side timing (SURVEY 8c item (2), 8d's
"within +-15 %" gate on the port's speed).

Runs only in the build container (the reference never travels to the GPU box); the committed record is
profiles/r03_oracle_vs_reference_full_size.json. The golden fixtures under tests/golden/ pin the oracle at h = 8 / 16;
this pins it at the size every GPU parity test of tests/test_parity_gpu.py relies on.

usage: python tools/check_oracle_full_size.py [--threads 8] [--frames 120] [--clips 2] [--out profiles/...json]
Layouts: C2 (MPHOI: H=2, O=4, N=26, classes (13, None), human segmentation = ones, object gates learned with recorded
Gumbel noise) and C1 (CAD-120: H=1, O=5, N=19, classes (10, 12), both segmentations given). N = 34 (C3) cannot run in the
unmodified reference (hard-coded split widths, vhoi/models.py:631-639)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('TWOG_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import cpu_ref  # noqa: E402
from vhoi.models import TGGCN as RefTGGCN  # noqa: E402

STAGE1 = dict(attention_style='v3', discrete_optimization_strategy='gs', filter_discrete_updates=False,
              message_humans_to_human=True, message_human_to_objects=True, message_objects_to_human=True,
              message_objects_to_object=True, message_geometry_to_objects=True, message_geometry_to_human=False,
              message_segment=True, message_type='v2', message_granularity='v1', message_aggregation='att',
              object_segment_update_strategy='ind', update_segment_threshold=0.5, bias=True, cat_level_states=0,
              share_level_mlps=0, add_segment_length=0, add_time_position=0, time_position_strategy='s',
              positional_encoding_style='e', discrete_networks_num_layers=1)


class GumbelRecorder:
    """Records the noise the reference draws (pyrutils/torch/distributions.py:16), in call order."""

    def __init__(self):
        self.drawn = []
        self._orig = torch.distributions.gumbel.Gumbel.sample

    def __enter__(self):
        rec = self

        def sample(self_, sample_shape=torch.Size()):
            g = rec._orig(self_, sample_shape)
            rec.drawn.append(g.clone())
            return g

        torch.distributions.gumbel.Gumbel.sample = sample
        return self

    def __exit__(self, *a):
        torch.distributions.gumbel.Gumbel.sample = self._orig


def inputs(bs, T, H, O, N, seed):
    g = torch.Generator().manual_seed(seed)
    vis = torch.relu(torch.randn(bs, T, H, 2048, generator=g))
    pos = torch.rand(bs, T, N, 2, generator=g)
    vel = torch.randn(bs, T, N, 2, generator=g) * 0.5
    geo = torch.cat([pos, vel], -1).reshape(bs, T, 1, 4 * N).expand(bs, T, H, 4 * N)
    x_human = torch.cat([vis, geo], -1).contiguous()
    x_objects = torch.relu(torch.randn(bs, T, O, 2048, generator=g))
    mask = torch.ones(bs, O)
    if bs > 1:
        mask[1, O - 2:] = 0.0
        x_objects[1, :, O - 2:] = 0.0
    return x_human, x_objects, mask, g


def one_layout(name, H, O, N, classes, both_given, bs, T, h, seed):
    cfg = dict(STAGE1, hidden_size=h, gcn_node=N)
    if H == 1:
        cfg['message_humans_to_human'] = False
    torch.manual_seed(seed)
    ref_model = RefTGGCN(input_size=(2048 + 4 * N, 2048), num_classes=classes, **cfg)
    ref_model.train()
    sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    x_human, x_objects, mask, g = inputs(bs, T, H, O, N, seed)
    kw = dict(steps_per_example=torch.full((bs,), float(T)))
    if both_given:
        kw['human_segmentation'] = (torch.rand(bs, T, H, generator=g) < 0.3).float()
        kw['objects_segmentation'] = (torch.rand(bs, T, O, generator=g) < 0.3).float()
    else:
        kw['human_segmentation'] = torch.ones(bs, T, H)
    # ---- the reference itself
    t0 = time.perf_counter()
    with GumbelRecorder() as rec:
        out_ref = ref_model(x_human, x_objects, mask, **kw)
    t_ref_fwd = time.perf_counter() - t0
    rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)) for i, o in enumerate(out_ref)]
    t0 = time.perf_counter()
    sum((o * r).sum() for o, r in zip(out_ref, rs) if o.requires_grad).backward()
    t_ref_bwd = time.perf_counter() - t0
    noise = torch.stack(rec.drawn) if rec.drawn else torch.zeros(0, bs, 2)
    # ---- the oracle on the same weights, inputs and noise
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
           for k, v in sd.items()}
    t0 = time.perf_counter()
    aux = {}
    out_or = cpu_ref.tggcn_forward(osd, cfg, x_human, x_objects, mask, training=True,
                                   gumbel_noise=noise if len(noise) else None, aux=aux, **kw)
    t_or_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    sum((o * r).sum() for o, r in zip(out_or, rs) if o.requires_grad).backward()
    t_or_bwd = time.perf_counter() - t0
    assert len(out_ref) == len(out_or)
    worst_out = 0.0
    for i, (a, b) in enumerate(zip(out_or, out_ref)):
        assert a.shape == b.shape, (name, i, a.shape, b.shape)
        worst_out = max(worst_out, (a.detach() - b.detach()).abs().max().item() / max(1.0, b.detach().abs().max().item()))
    # a gradient passes at 1e-4 of the tensor's scale (+ 1e-7 absolute: get_s.s2's bias has an exactly-zero true
    # gradient -- a per-row constant of the score cancels in the softmax -- so both sides hold rounding noise there)
    worst_grad, worst_name, n_cmp, worst_excess = 0.0, None, 0, 0.0
    for pname, p in ref_model.named_parameters():
        g_or = osd[pname].grad
        if p.grad is None:
            assert g_or is None or float(g_or.abs().max()) == 0.0, ('dead parameter with an oracle gradient', pname)
            continue
        assert g_or is not None, ('missing oracle gradient', pname)
        scale = max(p.grad.abs().max().item(), 1e-6)
        err = (g_or - p.grad).abs().max().item()
        n_cmp += 1
        worst_excess = max(worst_excess, err / (1e-4 * scale + 1e-7))
        if scale > 1e-5 and err / scale > worst_grad:
            worst_grad, worst_name = err / scale, pname
    bn_r, bn_o = ref_model.state_dict(), aux['bn_state']
    pre = 'geometry_embedding_gcn.joint_embed.cnn.0.bn.'
    bn_err = max((bn_o[k].detach() - bn_r[pre + k]).abs().max().item() for k in ('running_mean', 'running_var'))
    assert int(bn_o['num_batches_tracked']) == int(bn_r[pre + 'num_batches_tracked'])
    rec = dict(layout=name, H=H, O=O, N=N, h=h, bs=bs, T=T, classes=list(classes), both_segmentations_given=both_given,
               outputs=len(out_ref), worst_output_rel=worst_out, gradients_compared=n_cmp, worst_grad_rel=worst_grad,
               worst_grad_tensor=worst_name, worst_grad_over_tolerance=worst_excess, bn_running_stats_abs=bn_err,
               seconds=dict(reference_fwd=t_ref_fwd, reference_bwd=t_ref_bwd, oracle_fwd=t_or_fwd, oracle_bwd=t_or_bwd),
               clips_per_s_fwd_bwd=dict(reference=bs / (t_ref_fwd + t_ref_bwd), oracle=bs / (t_or_fwd + t_or_bwd)),
               oracle_over_reference_speed=(t_ref_fwd + t_ref_bwd) / (t_or_fwd + t_or_bwd))
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--frames', type=int, default=120)
    ap.add_argument('--clips', type=int, default=2)
    ap.add_argument('--hidden', type=int, default=512)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r03_oracle_vs_reference_full_size.json'))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    recs = [one_layout('C2 (MPHOI-72)', 2, 4, 26, (13, None), False, a.clips, a.frames, a.hidden, seed=9),
            one_layout('C1 (CAD-120)', 1, 5, 19, (10, 12), True, a.clips, a.frames, a.hidden, seed=21)]
    out = dict(tool='tools/check_oracle_full_size.py', torch=torch.__version__, threads=a.threads,
               host=os.uname().machine, tolerance=dict(outputs=1e-5, gradients='1e-4 * max|g| + 1e-7', bn_running_stats=1e-5), layouts=recs)
    ok = all(r['worst_output_rel'] < 1e-5 and r['worst_grad_over_tolerance'] < 1.0 and r['bn_running_stats_abs'] < 1e-5
             for r in recs)
    out['pass'] = ok
    json.dump(out, open(a.out, 'w'), indent=1)
    print('PASS' if ok else 'FAIL', a.out)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
