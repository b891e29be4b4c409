#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box). Weights come from the closed-form
generator in oracle/detgen.py, so the fixtures hold just inputs, outputs, (sampled) gradients and the Gumbel noise
the reference actually drew. Usage:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('TWOG_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import detgen  # noqa: E402
from vhoi.models import TGGCN, compute_attention_weights, compute_non_relational_message  # noqa: E402
from vhoi.models import reorder_hidden_states, filter_soft_decisions  # noqa: E402
from pyrutils.torch.models_gcn import Geo_gcn  # noqa: E402
from pyrutils.torch.models import build_mlp  # noqa: E402
import pyrutils.torch.distributions as ref_dist  # noqa: E402
from vhoi.losses import select_loss  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
GRAD_SAMPLE_LIMIT = 4096


def load_det(module, seed, gain=1.0):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    vals = detgen.fill_state_dict(shapes, seed=seed, gain=gain)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})


def sample_grad(g: torch.Tensor) -> np.ndarray:
    flat = g.detach().flatten().numpy()
    if flat.size <= GRAD_SAMPLE_LIMIT:
        return flat.copy()
    stride = flat.size // GRAD_SAMPLE_LIMIT
    return flat[::stride][:GRAD_SAMPLE_LIMIT].copy()


class GumbelRecorder:
    """Wraps torch.distributions.gumbel.Gumbel.sample to record the noise the reference draws."""

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


# ---------------------------------------------------------------------------------------------------------------
def g1_geo_gcn():
    """G1: Geo_gcn(N, 4, 128) train/eval, outputs + running stats + parameter grads."""
    out = {}
    for N in (19, 26, 30, 34):
        for mode in ('train', 'eval'):
            bs, T = 2, 5
            m = Geo_gcn(N, 4, 128)
            load_det(m, seed=100 + N)
            m.train(mode == 'train')
            x = torch.from_numpy(detgen.normal(f'g1.x.{N}', (bs, 4, N, T), std=1.0, seed=1))
            r = torch.from_numpy(detgen.normal(f'g1.r.{N}', (bs, 128, N, T), std=1.0, seed=2))
            y = m(x)
            (y * r).sum().backward()
            key = f'N{N}_{mode}'
            out[key + '_y'] = y.detach().numpy()
            for name, p in m.named_parameters():
                out[f'{key}_grad_{name}'] = p.grad.numpy().copy()
            bn = m.joint_embed.cnn[0].bn
            out[key + '_running_mean'] = bn.running_mean.numpy().copy()
            out[key + '_running_var'] = bn.running_var.numpy().copy()
            out[key + '_nbt'] = np.array(int(bn.num_batches_tracked))
    np.savez_compressed(os.path.join(OUT, 'g1_geo_gcn.npz'), **out)
    print('g1: ', len(out), 'arrays')


def g3_messages():
    """G3: compute_non_relational_message + compute_attention_weights for styles v1..v4 with masks, including a
    fully masked row (NaN -> 0 path, models.py:1750-1753)."""
    out = {}
    bs, S, d, hm = 4, 5, 12, 6
    q = torch.from_numpy(detgen.normal('g3.q', (bs, d), seed=3))
    keys = torch.from_numpy(detgen.normal('g3.k', (bs, S, d), seed=3))
    mask = torch.tensor([[1, 1, 1, 1, 1], [1, 0, 1, 0, 1], [0, 0, 0, 0, 0], [0, 0, 0, 1, 0]], dtype=torch.float32)
    out['q'], out['keys'], out['mask'] = q.numpy(), keys.numpy(), mask.numpy()
    msg = build_mlp([d, hm], ['relu'])
    load_det(msg, seed=31)
    out['msg_v1'] = compute_non_relational_message(q, keys, mask, 'v1', msg).detach().numpy()
    msg2 = build_mlp([2 * d, hm], ['relu'])
    load_det(msg2, seed=32)
    out['msg_v2'] = compute_non_relational_message(q, keys, mask, 'v2', msg2).detach().numpy()
    att1 = build_mlp([2 * d, 1], ['relu'])
    load_det(att1, seed=33)
    att4 = torch.nn.Bilinear(d, d, 1)
    load_det(att4, seed=34)
    for style, fn in (('v1', att1), ('v2', None), ('v3', None), ('v4', att4)):
        out['att_' + style] = compute_attention_weights(q, keys, mask, style, fn).detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'g3_messages.npz'), **out)
    print('g3: ', len(out), 'arrays')


def g5_reorder_filter():
    out = {}
    bs, T, d = 5, 11, 3
    hx = torch.from_numpy(detgen.normal('g5.hx', (bs, T, d), seed=5))
    ux = (torch.from_numpy(detgen.uniform01('g5.ux', (bs, T), seed=5)) > 0.6).float()
    ux[0, :] = 1.0
    ux[1, :] = 0.0  # no end flag at all
    ux[2, -1] = 0.0  # no trailing end flag
    ux[3, -1] = 1.0
    out['hx'], out['ux'] = hx.numpy(), ux.numpy()
    out['reordered'] = reorder_hidden_states(hx, ux).numpy()
    soft = torch.from_numpy(detgen.uniform01('g5.soft', (T, bs, 1), seed=6).astype(np.float32))
    for thr in (0.1, 0.5):
        f = filter_soft_decisions([s for s in soft], thr)
        out[f'filtered_{thr}'] = torch.stack(f, 0).numpy()
    out['soft'] = soft.numpy()
    np.savez_compressed(os.path.join(OUT, 'g5_reorder_filter.npz'), **out)
    print('g5: ', len(out), 'arrays')


# ---------------------------------------------------------------------------------------------------------------
STAGE1 = dict(add_segment_length=0, add_time_position=0, time_position_strategy='s', positional_encoding_style='e',
              attention_style='v3', bias=True, cat_level_states=0, discrete_networks_num_layers=1,
              discrete_optimization_strategy='gs', filter_discrete_updates=False, message_humans_to_human=True,
              message_human_to_objects=True, message_objects_to_human=True, message_objects_to_object=True,
              message_geometry_to_objects=True, message_geometry_to_human=False, message_segment=True,
              message_type='v2', message_granularity='v1', message_aggregation='att',
              object_segment_update_strategy='ind', share_level_mlps=0, update_segment_threshold=0.5)

CASES = {
    # name: (layout, H, O, N, hidden, bs, T, classes, cfg overrides, segmentation mode, seed)
    'c2_stage1': ('mphoi', 2, 4, 26, 16, 3, 7, (13, None), {}, 'human_ones', 11),
    'c2_stage2': ('mphoi', 2, 4, 26, 16, 2, 9, (13, None),
                  dict(filter_discrete_updates=True, update_segment_threshold=0.1), 'none', 12),
    'c1_stage1': ('cad120', 1, 5, 19, 16, 2, 8, (10, 12), dict(message_humans_to_human=False), 'both_given', 13),
    'c1_stage2': ('cad120', 1, 5, 19, 8, 2, 6, (10, 12),
                  dict(message_humans_to_human=False, filter_discrete_updates=True, update_segment_threshold=0.1),
                  'none', 14),
    'c5_stage1': ('bimanual', 2, 9, 30, 8, 2, 6, (14, None), {}, 'human_ones', 15),
    'c2_geo2human': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(message_geometry_to_human=True), 'human_ones', 16),
    'c2_dot_st': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None),
                  dict(attention_style='v2', discrete_optimization_strategy='st'), 'none', 17),
    # ---- round 2: the rest of the constructor's configuration surface (none is enabled by a shipped config)
    'c1_sah': ('cad120', 1, 5, 19, 8, 2, 6, (10, 12),
               dict(message_humans_to_human=False, object_segment_update_strategy='sah'), 'none', 21),
    'c1_coh': ('cad120', 1, 5, 19, 8, 2, 6, (10, 12),
               dict(message_humans_to_human=False, object_segment_update_strategy='coh'), 'none', 36),
    'c2_gate3': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(discrete_networks_num_layers=3), 'none', 23),
    'c2_nobias': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(bias=False), 'human_ones', 24),
    'c2_time_s': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None),
                  dict(add_time_position=1, time_position_strategy='s', positional_encoding_style='e'), 'human_ones', 25),
    'c2_time_u_periodic': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None),
                           dict(add_time_position=1, time_position_strategy='u', positional_encoding_style='p'),
                           'none', 26),
    'c2_seglen': ('mphoi', 2, 4, 26, 8, 2, 6, (13, None), dict(add_segment_length=1), 'none', 27),
    'c2_seglen_periodic': ('mphoi', 2, 4, 26, 8, 2, 6, (13, None),
                           dict(add_segment_length=1, positional_encoding_style='p'), 'human_ones', 28),
    'c2_concat': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(attention_style='v1'), 'human_ones', 29),
    'c2_general': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(attention_style='v4'), 'human_ones', 30),
    'c2_specific': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(message_granularity='v2'), 'human_ones', 31),
    'c2_relational': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(message_type='v1'), 'human_ones', 32),
    'c2_distance': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(), 'human_ones+dist', 33),
    'c2_ctor_defaults': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), 'defaults', 'human_ones', 34),
    # the same message / attention forms at the frame level only (message_segment off)
    'c2_concat_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(attention_style='v1', message_segment=False),
                    'human_ones', 42),
    'c2_general_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(attention_style='v4', message_segment=False),
                     'human_ones', 43),
    'c2_specific_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(message_granularity='v2', message_segment=False),
                      'human_ones', 44),
    'c2_specific_concat_mp_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None),
                                dict(message_granularity='v2', message_aggregation='mp', attention_style='v1',
                                     message_segment=False), 'none', 45),
    'c2_specific_general_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None),
                              dict(message_granularity='v2', attention_style='v4', message_segment=False,
                                   message_geometry_to_human=True), 'human_ones', 46),
    'c2_relational_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(message_type='v1', message_segment=False),
                        'human_ones', 47),
    'c2_distance_f': ('mphoi', 2, 4, 26, 8, 2, 5, (13, None), dict(message_segment=False), 'human_ones+dist', 48),
    'c1_relational_geo2h_f': ('cad120', 1, 5, 19, 8, 2, 5, (10, 12),
                              dict(message_humans_to_human=False, message_type='v1', message_geometry_to_human=True,
                                   message_segment=False), 'none', 49),
    'c1_relational_geo2h': ('cad120', 1, 5, 19, 8, 2, 5, (10, 12),
                            dict(message_humans_to_human=False, message_type='v1', message_geometry_to_human=True),
                            'none', 41),
}


def make_inputs(name, H, O, N, bs, T, seed):
    F_h = 2048 + 4 * N
    vis = np.maximum(detgen.normal(name + '.xh', (bs, T, H, 2048), seed=seed), 0.0)
    pos = detgen.uniform(name + '.pos', (bs, T, N, 2), 0.0, 1.0, seed=seed)
    vel = detgen.normal(name + '.vel', (bs, T, N, 2), std=0.5, seed=seed)
    geo = np.concatenate([pos, vel], axis=-1).reshape(bs, T, 1, 4 * N)
    geo = np.repeat(geo, H, axis=2)
    x_human = np.concatenate([vis, geo], axis=-1).astype(np.float32)
    assert x_human.shape[-1] == F_h
    x_objects = np.maximum(detgen.normal(name + '.xo', (bs, T, O, 2048), seed=seed), 0.0).astype(np.float32)
    mask = np.ones((bs, O), dtype=np.float32)
    mask[0, O - 1] = 0.0  # one virtual object
    if bs > 1 and O > 2:
        mask[1, O - 2:] = 0.0
    x_objects = x_objects * mask[:, None, :, None]
    return x_human, x_objects, mask


def g4_full(only=None):
    for name, (layout, H, O, N, hid, bs, T, classes, over, segmode, seed) in CASES.items():
        if only and name not in only:
            continue
        if over == 'defaults':   # the constructor's own defaults (vhoi/models.py:179-190)
            cfg = {}
        else:
            cfg = dict(STAGE1)
            cfg.update(over)
        cfg.update(hidden_size=hid, gcn_node=N)
        F_h = 2048 + 4 * N
        model = TGGCN(input_size=(F_h, 2048), num_classes=classes, **cfg)
        load_det(model, seed=seed, gain=1.6)
        model.train()
        x_human, x_objects, mask = make_inputs(name, H, O, N, bs, T, seed)
        kw = dict(x_human=torch.from_numpy(x_human), x_objects=torch.from_numpy(x_objects),
                  objects_mask=torch.from_numpy(mask), steps_per_example=torch.full((bs,), float(T)))
        seg_h = seg_o = None
        dists = {}
        if segmode.endswith('+dist'):   # centroid distances as the loader builds them; some exact zeros = "no sender"
            segmode = segmode[:-5]
            for key, shp in (('human_human_distances', (bs, T, H, H)), ('human_object_distances', (bs, T, H, O)),
                             ('object_object_distances', (bs, T, O, O))):
                d = detgen.uniform(name + '.' + key, shp, 0.05, 2.0, seed=seed).astype(np.float32)
                d[detgen.uniform01(name + '.z' + key, shp, seed=seed) < 0.15] = 0.0
                dists[key] = d
                kw[key] = torch.from_numpy(d)
        if segmode == 'human_ones':
            seg_h = np.ones((bs, T, H), dtype=np.float32)
        elif segmode == 'both_given':
            seg_h = (detgen.uniform01(name + '.segh', (bs, T, H), seed=seed) > 0.6).astype(np.float32)
            seg_o = (detgen.uniform01(name + '.sego', (bs, T, O), seed=seed) > 0.6).astype(np.float32)
            seg_h[:, -1] = 1.0
            seg_o[:, -1] = 1.0
        if seg_h is not None:
            kw['human_segmentation'] = torch.from_numpy(seg_h)
        if seg_o is not None:
            kw['objects_segmentation'] = torch.from_numpy(seg_o)
        torch.manual_seed(42)
        with GumbelRecorder() as rec:
            out = model(**kw)
        noise = torch.stack(rec.drawn, 0).numpy() if rec.drawn else np.zeros((0, bs, 2), np.float32)
        # scalar for backward: deterministic random projection of every output that carries grad
        loss = 0
        for i, o in enumerate(out):
            if o.requires_grad:
                r = torch.from_numpy(detgen.normal(f'{name}.r{i}', tuple(o.shape), seed=seed))
                loss = loss + (o * r).sum()
        backward_ok = True
        try:
            loss.backward()
        except RuntimeError as e:  # upstream bug: StraightThroughEstimator.backward returns 1 grad for 2 inputs
            backward_ok = False
            print(f'   [{name}] reference backward fails upstream: {str(e)[:90]}')
        save = dict(x_human=x_human, x_objects=x_objects, objects_mask=mask, gumbel_noise=noise,
                    loss=np.array(float(loss)))
        if seg_h is not None:
            save['human_segmentation'] = seg_h
        if seg_o is not None:
            save['objects_segmentation'] = seg_o
        save.update(dists)
        for i, o in enumerate(out):
            save[f'out{i}'] = o.detach().numpy()
        none_grads = []
        save['backward_ok'] = np.array(backward_ok)
        for pname, p in (model.named_parameters() if backward_ok else []):
            if p.grad is None:
                none_grads.append(pname)
            else:
                save['grad_' + pname] = sample_grad(p.grad)
        save['none_grads'] = np.array(none_grads)
        bn = model.geometry_embedding_gcn.joint_embed.cnn[0].bn
        save['bn_running_mean'] = bn.running_mean.numpy().copy()
        save['bn_running_var'] = bn.running_var.numpy().copy()
        if over == 'defaults':
            cfg = {k: getattr(model, k) for k in STAGE1 if hasattr(model, k)}
            cfg.update(hidden_size=hid, gcn_node=N, discrete_networks_num_layers=1, bias=True, share_level_mlps=False)
        meta = dict(layout=layout, H=H, O=O, N=N, hidden=hid, bs=bs, T=T, classes=list(classes), cfg=cfg,
                    segmode=segmode, seed=seed, gain=1.6,
                    state_dict_shapes={k: list(v.shape) for k, v in model.state_dict().items()})
        save['meta_json'] = np.array(__import__('json').dumps(meta))
        np.savez_compressed(os.path.join(OUT, f'g4_{name}.npz'), **save)
        soft = [o for o in out[:4] if o.dim() == 3]
        thr = cfg['update_segment_threshold']
        margins = [float((o.detach() - thr).abs().min()) for o in soft]
        print(f'g4 {name}: outs={len(out)} noise={noise.shape} none_grads={len(none_grads)} '
              f'min|soft-thr|={min(margins):.4f} loss={float(loss):.5f}')


def g7_losses():
    """G7: vhoi.losses.select_loss list on fixed log-prob outputs/targets (pins the fwd+bwd scalar of bench.py)."""
    class Cfg(dict):
        def get(self, k, default_value=None, **kw):
            return dict.get(self, k, default_value if default_value is not None else kw.get('default'))

    out = {}
    for ds, n_out, classes in (('mphoi', 6, 13), ('cad120', 12, 10)):
        bs, T, E = 2, 6, 2
        misc = dict(anticipation_loss_weight=1.0, budget_loss=dict(add=True, human_weight=0.5, object_weight=0.25),
                    first_level_loss_weight=0.3, segmentation_loss=dict(add=True, pretrain=False, weight=0.7))
        crit, names = select_loss('2G-GCN', 'multiple', ds, Cfg(misc=misc))
        outs, tgts = [], []
        n_b = 2 if ds == 'cad120' else 1
        for i in range(n_out):
            if i < 2 * n_b:
                o = torch.from_numpy(detgen.uniform(f'g7.{ds}.o{i}', (bs, T, E), 0.05, 0.95, seed=7))
                t = (torch.from_numpy(detgen.uniform01(f'g7.{ds}.t{i}', (bs, T, E), seed=7)) > 0.5).float()
                t[0, -2:] = -1.0
            else:
                o = torch.log_softmax(torch.from_numpy(detgen.normal(f'g7.{ds}.o{i}', (bs, classes, T, E), seed=7)), 1)
                t = torch.from_numpy((detgen.uniform01(f'g7.{ds}.t{i}', (bs, T, E), seed=7) * classes).astype(np.int64))
                t[0, -2:] = -1
            outs.append(o)
            tgts.append(t)
            out[f'{ds}_o{i}'], out[f'{ds}_t{i}'] = o.numpy(), t.numpy()
        losses = crit(outs, tgts)
        out[f'{ds}_losses'] = np.array([float(v) for v in losses], dtype=np.float64)
        out[f'{ds}_names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'g7_losses.npz'), **out)
    print('g7: ', len(out), 'arrays')


def _raw_videos(kind, seed):
    """Synthetic raw per-video arrays in the layout the reference's readers produce (small feature width)."""
    rng = np.random.RandomState(seed)
    vids = []
    F = 16
    if kind in ('mphoi', 'bimanual'):
        J = 32 if kind == 'mphoi' else 21
        n_obj_max = 4 if kind == 'mphoi' else 9
        keys = ('Human1', 'Human2') if kind == 'mphoi' else ('left_hand', 'right_hand')
        for L, n in ((20, n_obj_max), (14, n_obj_max - 1), (17, 2)):
            gt = {}
            for k in keys:
                y, cur = [], 0
                while len(y) < L:
                    y += [int(rng.randint(0, 5))] * int(rng.randint(2, 6))
                gt[k] = y[:L]
            vids.append([rng.randn(L, F).astype(np.float32), rng.randn(L, F).astype(np.float32),
                         rng.randn(L, n, F).astype(np.float32), gt,
                         rng.rand(L, 4) * 1000, rng.rand(L, 4) * 1000, rng.rand(L, n, 4) * 1000,
                         rng.rand(L, J, 2) * 1000, rng.rand(L, J, 2) * 1000])
    else:
        from vhoi.cad120classes import CAD120VideoSegment
        for L, n in ((21, 5), (15, 3), (18, 4)):
            segs, start = [], 1
            while start <= L:
                end = min(L, start + int(rng.randint(2, 6)))
                s = CAD120VideoSegment()
                s.start_frame, s.end_frame = start, end
                s.subactivity = int(rng.randint(1, 11))
                s.object_affordance = {o + 1: int(rng.randint(1, 13)) for o in range(n)}
                segs.append(s)
                start = end + 1
            for a, b in zip(segs[:-1], segs[1:]):
                a.next_subactivity = b.subactivity
                a.next_object_affordance = dict(b.object_affordance)
            vids.append([rng.randn(L, F).astype(np.float32), rng.randn(L, n, F).astype(np.float32),
                         rng.rand(L, 4) * 400, rng.rand(L, n, 4) * 400, rng.rand(L, 9, 2) * 300, segs])
    return vids


def g6_batching():
    """G6: the reference's create_data_loader on synthetic raw videos (mphoi / bimanual / cad120), the fetcher's device
    placement decisions and the feeder's forward kwargs."""
    import types
    sys.modules.setdefault('zarr', types.ModuleType('zarr'))
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = object
    sys.modules.setdefault('torch.utils.tensorboard', tb)
    import vhoi.data_loading as ref_dl
    out = {}
    for kind in ('mphoi', 'bimanual', 'cad120'):
        for sigma, test_data in ((0.0, False), (2.0, False), (0.0, True)):
            vids = _raw_videos(kind, seed=60)
            loader, _, _ = ref_dl.create_data_loader(vids, '2G-GCN', 'multiple', kind, batch_size=2, shuffle=False,
                                                     sigma=sigma, downsampling=3, test_data=test_data)
            for i, t in enumerate(loader.dataset.tensors):
                out[f'{kind}_s{sigma}_t{int(test_data)}_{i}'] = t.numpy()
        loader, _, _ = ref_dl.create_data_loader(_raw_videos(kind, seed=60), '2G-GCN', 'multiple', kind, batch_size=2,
                                                 shuffle=False, downsampling=3)
        batch = next(iter(loader))

        class Rec:
            def __call__(self, **kw):
                self.kw = kw
                return 'out'

        for tag, kwargs in (('plain', dict(dataset_name=kind, impose_segmentation_pattern=1)),
                            ('input', dict(dataset_name=kind, input_human_segmentation=True,
                                           input_object_segmentation=True, make_attention_distance_based=True))):
            data, targets = ref_dl.gcn_fetcher(batch, 'cpu', **kwargs)
            rec = Rec()
            ref_dl.gcn_forward(rec, data, **kwargs)
            out[f'{kind}_{tag}_n_targets'] = np.array(len(targets))
            for k, v in rec.kw.items():
                if torch.is_tensor(v):
                    out[f'{kind}_{tag}_kw_{k}'] = v.numpy()
                else:
                    out[f'{kind}_{tag}_kwnone_{k}'] = np.array(v is None or v is False)
    np.savez_compressed(os.path.join(OUT, 'g6_batching.npz'), **out)
    print('g6: ', len(out), 'arrays')


def g8_postprocess():
    """G8: inference post-processing of predict.py (:64-70 repeat_interleave + match_shape, :195-201 argmax) and the
    segmental metric pyrutils.metrics.f1_at_k (:68-81) on seeded label sequences with ignored padding."""
    from pyrutils.metrics import f1_at_k
    # predict.py imports omegaconf / sklearn at module level (absent here); match_shape is self-contained: load it alone
    src = open(os.path.join(REF, 'predict.py')).read()
    start = src.index('def match_shape(out, tgt):')
    end = src.index('def match_att_shape')
    ns = {'torch': torch}
    exec(compile(src[start:end], 'predict_match_shape', 'exec'), ns)
    match_shape = ns['match_shape']
    out = {}
    rng = np.random.RandomState(8)
    cases = []
    for ci, (bs, C, T, E, ds, T_tgt) in enumerate(((3, 13, 10, 2, 1, 10), (2, 10, 7, 5, 3, 19), (2, 12, 6, 3, 4, 26),
                                                   (1, 4, 5, 1, 2, 7))):
        logp = torch.log_softmax(torch.from_numpy(rng.randn(bs, C, T, E).astype(np.float32)), 1)
        logp[0, 1, 0, 0] = logp[0, 2, 0, 0] = logp[0].max() + 1.0     # a tie: np.argmax takes the first index
        tgt = torch.zeros(bs, T_tgt, E, dtype=torch.int64)
        o = logp
        if ds > 1:
            o = torch.repeat_interleave(o, repeats=ds, dim=-2)
            o = match_shape(o, tgt)
        labels = np.argmax(o.numpy(), axis=1)
        out[f'pl{ci}_logp'], out[f'pl{ci}_labels'] = logp.numpy(), labels.astype(np.int64)
        out[f'pl{ci}_cfg'] = np.array([ds, T_tgt if ds > 1 else T])
        cases.append(ci)
    out['pl_cases'] = np.array(cases)
    # F1@k: sequences built from runs, predictions = noisy copies; trailing padding -1 and a fully padded row
    for fi, (n_seq, n_steps, ncls) in enumerate(((6, 40, 5), (4, 25, 3), (3, 8, 2))):
        yt = np.zeros((n_seq, n_steps), dtype=np.int64)
        yp = np.zeros((n_seq, n_steps), dtype=np.int64)
        for r in range(n_seq):
            seq, pred = [], []
            while len(seq) < n_steps:
                run = int(rng.randint(1, 7))
                lab = int(rng.randint(0, ncls + 1))          # label ncls is "ignored class" (>= num_classes)
                seq += [lab] * run
                shift = int(rng.randint(-2, 3))
                plab = lab if rng.rand() < 0.7 else int(rng.randint(0, ncls + 1))
                pred += [plab] * max(1, run + shift)
            yt[r] = seq[:n_steps]
            yp[r] = (pred + [pred[-1]] * n_steps)[:n_steps]
            pad = int(rng.randint(0, 6))
            if pad:
                yt[r, n_steps - pad:] = -1
        yt[-1, :] = -1 if fi == 0 else yt[-1, :]
        out[f'f1_{fi}_true'], out[f'f1_{fi}_pred'] = yt, yp
        vals = [f1_at_k(yt, yp, ncls, overlap=ov, ignore_value=-1.0)
                for ov in (0.1, 0.25, 0.5)]
        out[f'f1_{fi}_values'] = np.array(vals, dtype=np.float64)
        out[f'f1_{fi}_ncls'] = np.array(ncls)
    np.savez_compressed(os.path.join(OUT, 'g8_postprocess.npz'), **out)
    print('g8: ', len(out), 'arrays')


def g11_segmentation_helpers():
    """G11: the segmentation helpers of vhoi/data_loading.py on seeded label matrices with missing (-1) frames, bit for
    bit: smooth_segmentation (:544-559; result AND what it leaves in the caller's array), segmentation_from_output_class
    (:885-896), ignore_last_step_end_flag (:524-533)."""
    import types
    sys.modules.setdefault('zarr', types.ModuleType('zarr'))
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = object
    sys.modules.setdefault('torch.utils.tensorboard', tb)
    import vhoi.data_loading as ref_dl
    rng = np.random.RandomState(11)
    out = {}
    for ci, (n, steps) in enumerate(((5, 40), (3, 17), (2, 6))):
        labels = np.repeat(rng.randint(0, 4, size=(n, steps // 3 + 1)), 3, axis=1)[:, :steps].astype(np.int64)
        labels[0, steps - 4:] = -1                       # trailing padding
        labels[-1, 1:3] = -1                             # missing frames inside a clip
        out[f'c{ci}_labels'] = labels
        for style in ('input', 'output'):
            seg = ref_dl.segmentation_from_output_class(labels.copy(), segmentation_type=style)
            out[f'c{ci}_seg_{style}'] = np.asarray(seg)
        seg = np.asarray(out[f'c{ci}_seg_output'], dtype=np.float32)
        for sigma in (0.0, 1.0, 2.0, 3.5):
            arg = seg.copy()
            res = ref_dl.smooth_segmentation(arg, sigma)
            out[f'c{ci}_smooth_{sigma}'] = np.asarray(res)
            out[f'c{ci}_smooth_{sigma}_arg_after'] = arg
        out[f'c{ci}_ignore_last'] = ref_dl.ignore_last_step_end_flag(np.asarray(out[f'c{ci}_seg_input'], dtype=np.float32).copy())
    np.savez_compressed(os.path.join(OUT, 'g11_segmentation_helpers.npz'), **out)
    print('g11:', len(out), 'arrays')


G12_PARAMS = ['human_recognition_mlp.0.weight', 'objects_to_human_message_mlp.0.weight', 'human_segment_rnn_fcell.weight_hh',
              'object_bd_rnn.weight_ih_l0', 'geometry_embedding_gcn.weight', 'update_human_segment_mlp.0.weight',
              'objects_to_object_segment_message_mlp.0.bias', 'geometry_embedding_mlp.2.weight']
G12_MISC = dict(anticipation_loss_weight=1.0, budget_loss=dict(add=True, human_weight=0.5, object_weight=0.25),
                first_level_loss_weight=0.3, segmentation_loss=dict(add=True, pretrain=False, weight=0.7))
G12 = dict(name='g12', H=2, O=4, N=26, hid=16, bs=3, T=8, classes=(13, None), seed=51, gain=1.6, steps=5, lr=1e-4)


def g12_training_trajectory():
    """G12: the reference's TRAINING STEP composed five times -- reference TGGCN (MPHOI layout, every gate learned, train
    mode) + vhoi.losses.select_loss (every term weighted) + torch.optim.Adam(lr=1e-4), in the order of
    pyrutils/torch/train_utils.py:143-154 (zero_grad, forward, criterion(reduction='mean'), sum, backward, step); train.py:38-46
    builds exactly this optimizer and criterion. A fresh batch and fresh Gumbel noise per step (drawn from the default CPU
    generator, recorded). Stored: the noise, per-step loss lists, BatchNorm running statistics and eight parameters after
    step 5 (inputs, targets and initial weights are closed-form: oracle/detgen.py)."""
    class Cfg(dict):
        def get(self, k, default_value=None, **kw):
            return dict.get(self, k, default_value if default_value is not None else kw.get('default'))

    c = G12
    cfg = dict(STAGE1)
    cfg.update(hidden_size=c['hid'], gcn_node=c['N'])
    model = TGGCN(input_size=(2048 + 4 * c['N'], 2048), num_classes=c['classes'], **cfg)
    load_det(model, seed=c['seed'], gain=c['gain'])
    model.train()
    crit, names = select_loss('2G-GCN', 'multiple', 'mphoi', Cfg(misc=G12_MISC))
    opt = torch.optim.Adam(model.parameters(), lr=c['lr'])
    save = {}
    init = {n: p.detach().clone() for n, p in model.named_parameters()}
    torch.manual_seed(42)
    losses_all, hard_all = [], []
    for step in range(c['steps']):
        # inputs and targets: the closed-form generators the tests use (tests/helpers.py), so the fixture need not hold them
        from tests.helpers import g12_step_batch
        kw, target = g12_step_batch(dict(c, classes=list(c['classes'])), step)
        chk = make_inputs(f'g12.s{step}', c['H'], c['O'], c['N'], c['bs'], c['T'], c['seed'])
        assert np.array_equal(chk[0], kw['x_human'].numpy()) and np.array_equal(chk[1], kw['x_objects'].numpy())
        with GumbelRecorder() as rec:
            opt.zero_grad()
            out = model(**kw)
            losses = crit(out, target, reduction='mean')
            loss = sum(losses)
            loss.backward()
            opt.step()
        save[f'noise{step}'] = torch.stack(rec.drawn, 0).numpy()
        losses_all.append([float(v) for v in losses])
        hard_all.append(out[0].detach().numpy().copy())
        soft = out[1].detach()
        print(f'g12 step {step}: loss {float(loss):.6f}  terms {[round(float(v), 5) for v in losses]}  min|soft-0.5| '
              f'{float((soft - 0.5).abs().min()):.4f}')
    save['losses'] = np.array(losses_all, dtype=np.float64)
    save['hard_gates'] = np.stack(hard_all, 0)
    save['loss_names'] = np.array(names)
    bn = model.geometry_embedding_gcn.joint_embed.cnn[0].bn
    save['bn_running_mean'], save['bn_running_var'] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
    save['bn_num_batches_tracked'] = np.array(int(bn.num_batches_tracked))
    P = dict(model.named_parameters())
    for n in G12_PARAMS:
        save['final_' + n] = sample_grad(P[n])
        save['delta_' + n] = sample_grad(P[n].detach() - init[n])
    # every parameter's largest move and whether it moved at all (dead parameters must stay put)
    save['moved'] = np.array([n for n, p in P.items() if float((p.detach() - init[n]).abs().max()) > 0])
    meta = dict(cfg=cfg, misc=G12_MISC, params=G12_PARAMS, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items()},
                state_dict_shapes={k: list(v.shape) for k, v in model.state_dict().items()})
    save['meta_json'] = np.array(__import__('json').dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'g12_training_trajectory.npz'), **save)
    print('g12:', len(save), 'arrays')


if __name__ == '__main__':
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'g11':
        g11_segmentation_helpers()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g12':
        g12_training_trajectory()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'g4':   # python tools/make_golden.py g4 [case ...]
        g4_full(only=set(sys.argv[2:]) or None)
        sys.exit(0)
    g1_geo_gcn()
    g3_messages()
    g5_reorder_filter()
    g4_full()
    g7_losses()
    g6_batching()
    g8_postprocess()
    g11_segmentation_helpers()
    g12_training_trajectory()
