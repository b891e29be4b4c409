#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: the shipped HIP path vs the CPU oracle (oracle/cpu_ref.py) over random small
layouts -- clips, frames, humans, objects, graph nodes, hidden width, object masks, gate semantics (given / learned
segmentation, local-maximum filter), message switches, attention style, train / eval mode -- outputs at 1e-4 relative to
the fp32 oracle, parameter gradients at 2e-5 of each tensor's scale against the oracle run in fp64.
usage: python3 tools/parity_fuzz.py [n_cases] [seed] [first_case]     (writes gpurun_out/parity_fuzz.json; seeds >= 100
also draw large layouts)"""
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd.hostcpu import limit_host_threads  # noqa: E402
limit_host_threads()
from twog_gcn_amd import kernels  # noqa: E402
from twog_gcn_amd.models import TGGCN  # noqa: E402
from oracle import cpu_ref  # noqa: E402

DEV = 'cuda:0'
BASE = dict(attention_style='v3', discrete_optimization_strategy='gs', filter_discrete_updates=False,
            message_humans_to_human=True, message_human_to_objects=True, message_objects_to_human=True,
            message_objects_to_object=True, message_geometry_to_objects=True, message_geometry_to_human=False,
            message_segment=True, message_type='v2', message_granularity='v1', message_aggregation='att',
            object_segment_update_strategy='ind', update_segment_threshold=0.5)


def _stage_branch(name):
    """(forward stage, branch) of a parameter: which tensors lie upstream of which (ReLU-boundary rule below).
    Branch g / h / o = geometry / human / object stream before the streams mix in the segment level (stage >= 6)."""
    if name.startswith('geometry_embedding_gcn.'):   # inside the GCN: BatchNorm -> 1x1 conv -> 1x1 conv -> attention, weight
        for sub, st in (('joint_embed.cnn.0.', 0.0), ('joint_embed.cnn.1.', 0.1), ('joint_embed.cnn.3.', 0.2)):
            if sub in name:
                return st, 'g'
        return 0.3, 'g'
    if name.startswith('geometry_embedding_mlp.0'):
        return 1, 'g'
    if name.startswith('geometry_embedding_mlp.2'):
        return 2, 'g'
    if name.startswith('human_embedding_mlp'):
        return 2, 'h'
    if name.startswith('object_embedding_mlp'):
        return 2, 'o'
    for b, pre in (('g', 'geometry'), ('h', 'human'), ('o', 'object')):
        if name.startswith(pre + '_bd_rnn'):
            return 3, b
        if name.startswith(pre + '_bd_embedding_mlp'):
            return 4, b
    if '_segment_' in name or name.startswith('update_') or name.startswith('time_position') or \
            name.startswith('segment_length'):
        return 6, None
    if 'recognition_mlp' in name or 'prediction_mlp' in name:
        return 7, None
    if '_message_' in name or '_relation_mlp' in name:   # frame-level message / attention MLPs: sender's stream
        sender = name.split('_to_')[0]
        return 5, {'humans': 'h', 'human': 'h', 'objects': 'o', 'geometry': 'g'}.get(sender)
    return 6, None


def _is_owner_or_upstream(t, owner):
    """True when parameter `t` is `owner` or feeds it: only those gradients may move when one ReLU unit of `owner`'s
    layer sits on the other side of zero (everything downstream and every parallel stream keeps its gradient)."""
    if t == owner or t.rsplit('.', 1)[0] == owner.rsplit('.', 1)[0]:   # the layer's weight and bias
        return True
    (st, bt), (so, bo) = _stage_branch(t), _stage_branch(owner)
    if so >= 6:           # the streams have mixed (and the segment level is recurrent): everything up to it
        return st <= 6
    if so == 5 and st == 5:   # frame-level message MLPs do not feed each other
        return False
    return st < so and (bt == bo or bt is None or bo is None)


SMALL_TENSOR = 16   # elements
SMALL_TENSOR_NOISE_FACTOR = 4.0   # x the largest of eight samples of the fp32 oracle's own rounding noise


def _fp32_noise_of_small_tensors(names, sd, cfg, x_human, x_objects, mask, kw, training, noise_in, recorded, rs, g64, samples=8):
    """{name: max over `samples` re-runs of the fp32 oracle of |gradient - fp64 gradient|}; every re-run on parameters
    and features multiplied by 1 + d, |d| <= 2^-23 (one unit in the last place), replaying the recorded hard decisions."""
    worst = {n: 0.0 for n in names}
    for k in range(samples):
        g = torch.Generator().manual_seed(7700 + k)

        def moved(v):
            return v * (1.0 + 2.0 ** -23 * (2.0 * torch.rand(v.shape, generator=g) - 1.0))

        osd = {n: (moved(v.detach()).requires_grad_(True) if v.is_floating_point() and 'running' not in n else v.clone())
               for n, v in sd.items()}
        ref = cpu_ref.tggcn_forward(osd, dict(cfg), moved(x_human), moved(x_objects), mask, training=training,
                                    gumbel_noise=noise_in, decisions=cpu_ref.DecisionTape(recorded), **kw)
        sum((o * r).sum() for o, r in zip(ref, rs) if o.requires_grad).backward()
        for n in names:
            worst[n] = max(worst[n], (osd[n].grad.double() - g64[n]).abs().max().item())
    return worst


def _decision_margin(aux, ref, cfg, given_seg, cad):
    """Smallest distance of a LEARNED soft gate of the oracle run to what its hard decision is compared with: the
    threshold and, with the local-maximum filter (vhoi/models.py:1637-1664), the neighbouring time steps."""
    thr, filt = cfg['update_segment_threshold'], cfg['filter_discrete_updates']
    series = []   # (T, n) tensors of soft values
    if not given_seg:
        soft_h = ref[2 if cad else 1].detach()                       # y_hss (bs, T, H)
        series.append(soft_h.permute(1, 0, 2).reshape(soft_h.shape[1], -1))
    for per in aux.get('ux_oss', []) or []:                          # per object: list over time of (bs, 1)
        if isinstance(per, (list, tuple)) and len(per) and torch.is_tensor(per[0]):
            series.append(torch.stack([x.detach().reshape(-1) for x in per]))
    best = float('inf')
    for S in series:
        if not ((S != 0) & (S != 1)).any():   # a GIVEN segmentation (exact 0 / 1 on both sides): no rounding involved
            continue
        best = min(best, float((S - thr).abs().min()))
        if filt and S.shape[0] > 1:
            best = min(best, float((S[1:] - S[:-1]).abs().min()))
    return best


def one_case(rng, idx, dev=DEV, dry=False, run_seed=0):
    H = rng.choice([1, 2, 2])
    cfg = dict(BASE)
    bs, T = rng.randint(1, 5), rng.randint(1, 9)
    if run_seed >= 100:
        # sweeps with seed >= 100 also draw LARGE layouts (several 64-row tiles per entity type, chains of up to 18 steps)
        # for one case in eight, from a generator of their own: the main sequence -- and with it every case of the
        # recorded sweeps (seeds < 100) -- is unchanged
        rng2 = random.Random((run_seed << 20) + idx)
        if rng2.random() < 0.125:
            bs, T = rng2.randint(6, 14), rng2.randint(10, 18)
    O, N = rng.randint(1, 12), rng.choice([19, 26, 30, 34, 21, 40])
    if run_seed >= 1000:
        # sweeps with seed >= 1000 also draw clips with MORE entities than the tuned attention kernel holds (H > 4 or O > 12:
        # served by the general single-relation kernels since round 6) for one case in eight, again from a generator of
        # their own (the main sequence, and with it every recorded case, is unchanged)
        rng3 = random.Random((run_seed << 21) + idx)
        if rng3.random() < 0.125:
            if rng3.random() < 0.5:
                O = rng3.randint(13, 16)
            else:
                H = rng3.randint(3, 5)
            N = rng3.choice([40, 46, 50])
    h = rng.choice([16, 32, 48, 64, 80, 16, 32, 48, 64, 80, 256, 128, 64])   # 256: four column tiles per GEMM problem; 64 / 128 / 256: the persistent segment launches (round 5)
    if H == 1:
        cfg['message_humans_to_human'] = False
    if O == 1:
        cfg['message_objects_to_object'] = False   # the reference stacks an empty sender list otherwise
    cfg['message_geometry_to_human'] = rng.random() < 0.3
    cfg['message_geometry_to_objects'] = rng.random() < 0.8
    cfg['message_segment'] = rng.random() < 0.85
    cfg['filter_discrete_updates'] = rng.random() < 0.3
    if rng.random() < 0.2:
        cfg['message_human_to_objects'] = False
    if rng.random() < 0.2:
        cfg['message_objects_to_human'] = False
    cfg['attention_style'] = rng.choice(['v3', 'v3', 'v2'])
    cfg['message_aggregation'] = rng.choice(['att', 'att', 'att', 'mp'])
    cfg['share_level_mlps'] = rng.random() < 0.2
    cfg['cat_level_states'] = rng.random() < 0.2
    cfg['discrete_optimization_strategy'] = rng.choice(['gs', 'gs', 'gs', 'st'])
    cfg['update_segment_threshold'] = rng.choice([0.5, 0.5, 0.3, 0.7])
    cfg['bias'] = rng.random() < 0.85
    # round 2: the rest of the constructor surface
    cfg['discrete_networks_num_layers'] = rng.choice([1, 1, 1, 2, 3])
    if H == 1:
        cfg['object_segment_update_strategy'] = rng.choice(['ind', 'ind', 'sah', 'coh'])
    if rng.random() < 0.25:
        cfg['add_time_position'] = 1
        cfg['time_position_strategy'] = rng.choice(['s', 'u'])
    if rng.random() < 0.2:
        cfg['add_segment_length'] = 1
    cfg['positional_encoding_style'] = rng.choice(['e', 'p'])
    if h % 2:
        cfg['positional_encoding_style'] = 'e'
    general = rng.random() < 0.3
    if general:   # relational / receiver-specific messages, concat / bilinear attention (both levels)
        cfg['message_type'] = rng.choice(['v1', 'v2', 'v2'])
        cfg['message_granularity'] = rng.choice(['v1', 'v2'])
        cfg['attention_style'] = rng.choice(['v1', 'v4', 'v3'])
    use_dist = rng.random() < 0.2
    cad = H == 1 and rng.random() < 0.5
    classes = (10, 12) if cad else (13, None)
    training = rng.random() < 0.8
    given_seg = rng.random() < 0.5
    desc = dict(idx=idx, bs=bs, T=T, H=H, O=O, N=N, h=h, classes=classes, training=training, given_seg=given_seg,
                geo2h=cfg['message_geometry_to_human'], geo2o=cfg['message_geometry_to_objects'],
                seg_msg=cfg['message_segment'], filt=cfg['filter_discrete_updates'], h2o=cfg['message_human_to_objects'],
                o2h=cfg['message_objects_to_human'], att=cfg['attention_style'], agg=cfg['message_aggregation'], share=cfg['share_level_mlps'], cat=cfg['cat_level_states'], strat=cfg['discrete_optimization_strategy'],
                thr=cfg['update_segment_threshold'], bias=cfg['bias'], layers=cfg['discrete_networks_num_layers'],
                ostrat=cfg['object_segment_update_strategy'], time=cfg.get('add_time_position', 0) and cfg['time_position_strategy'],
                seglen=cfg.get('add_segment_length', 0), pos=cfg['positional_encoding_style'], mtype=cfg['message_type'],
                gran=cfg['message_granularity'], dist=use_dist)
    seed = rng.randint(0, 10 ** 6)
    if dry:   # only advance the case generator (tools/parity_fuzz.py N SEED FIRST: replay from case FIRST)
        if rng.random() < 0.15:
            rng.randrange(bs)
        return desc
    torch.manual_seed(seed)
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=classes, hidden_size=h, gcn_node=N, **cfg)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed + 1)
    x_human = torch.rand(bs, T, H, 2048 + 4 * N, generator=g)
    x_objects = torch.rand(bs, T, O, 2048, generator=g)
    mask = (torch.rand(bs, O, generator=g) < 0.75).float()
    if rng.random() < 0.15:
        mask[rng.randrange(bs)] = 0            # a clip without any real object
    x_objects = x_objects * mask[:, None, :, None]
    kw = {}
    if given_seg:
        kw['human_segmentation'] = (torch.rand(bs, T, H, generator=g) < 0.6).float()
        if cad:
            kw['objects_segmentation'] = (torch.rand(bs, T, O, generator=g) < 0.6).float()
    kw['steps_per_example'] = torch.full((bs,), float(T)) - torch.randint(0, max(1, T // 2), (bs,), generator=g).float()
    if use_dist:
        def dd(*shape):
            d = torch.rand(*shape, generator=g) * 2 + 0.05
            d[torch.rand(*shape, generator=g) < 0.15] = 0.0
            return d
        kw.update(human_human_distances=dd(bs, T, H, H), human_object_distances=dd(bs, T, H, O),
                  object_object_distances=dd(bs, T, O, O))
    sah_alias = cfg['object_segment_update_strategy'] == 'sah' and H == 1
    n_gated = (0 if given_seg else H) + (0 if ((given_seg and cad) or sah_alias) else O)
    noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * max(n_gated, 1), bs, 2))
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
           for k, v in sd.items()}
    ref_aux = {}
    tape = cpu_ref.DecisionTape()   # the hard decisions of the fp32 oracle run, replayed by the fp64 run below
    ref = cpu_ref.tggcn_forward(osd, dict(m.cfg), x_human, x_objects, mask, training=training, gumbel_noise=noise,
                                aux=ref_aux, decisions=tape, **kw)
    if os.environ.get('TWOG_FUZZ_ORACLE_ONLY'):   # dry run of the case generator + oracle (no GPU)
        return dict(desc, worst_output_rel=0.0, worst_grad_rel=0.0)
    m = m.to(dev)
    m.train(training)
    m._gumbel_noise_override = noise
    try:
        out = m(x_human.to(dev), x_objects.to(dev), mask.to(dev), **{k: v.to(dev) for k, v in kw.items()})
    except NotImplementedError as e:   # a configuration the gfx950 path declares unsupported (loudly): not a parity case
        raise AssertionError(f'the generator only draws supported configurations: {e}')
    assert len(out) == len(ref)
    worst_out, flipped = 0.0, None
    for i, (o, r) in enumerate(zip(out, ref)):
        got, want = o.detach().cpu(), r.detach()
        assert got.shape == want.shape, (i, got.shape, want.shape)
        assert not torch.isnan(got).any(), ('nan', i)
        err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        if err >= 1e-4 and flipped is None:
            # A hard decision on the rounding boundary: a learned soft gate within a few ulp of the threshold or -- with
            # the local-maximum filter -- of its neighbour in time (seed 43 #285: two consecutive soft values 5.96e-8 = one
            # ulp apart decide which of the two frames ends a segment). The reference's own result then depends on the
            # order of its fp32 sums. Accepted only if the oracle's soft gates show such a margin; recorded with the
            # margin, and the gradients are not judged.
            margin = _decision_margin(ref_aux, ref, cfg, given_seg, cad)
            flipped = dict(output=i, kernels_vs_oracle=err, smallest_decision_margin=margin)
            assert margin < 4e-7, ('output', i, err, 'smallest decision margin', margin)
        if flipped is not None:
            assert err < 0.5, ('output', i, err, flipped)   # a flipped segment end moves the segment level, nothing explodes
            continue
        worst_out = max(worst_out, err)
        assert err < 1e-4, ('output', i, err)
    if flipped is not None:
        desc.update(decision_on_rounding_boundary=flipped, worst_output_rel=worst_out, worst_grad_rel=0.0,
                    worst_grad_rel_ill_conditioned=0.0, tensors_judged_by_conditioning=0, grad_ref='not judged')
        return desc
    worst_g, worst_cond, n_cond, grad_ref, worst_tiny = 0.0, 0.0, 0, 'none', 0.0
    st_learned = cfg['discrete_optimization_strategy'] == 'st' and n_gated > 0
    if training and not st_learned:   # 'st' with learned gates: the reference's backward raises (upstream bug), forward only
        # Gradient reference: the oracle run in fp64. The fp32 CPU restatement is itself off by up to 1e-2 on the
        # BatchNorm-conditioned GCN parameters at these tiny batches (tools/parity_fuzz_diag.py), so it cannot judge the
        # kernels at 5e-4; fp64 can. The fp64 run REPLAYS the fp32 run's hard decisions (oracle/cpu_ref.py::DecisionTape:
        # `soft > threshold`, the comparisons of the local-maximum filter), so both follow the same discrete path even when
        # a soft gate lies within fp32 rounding of its threshold -- there is no weaker fallback gate any more (VERDICT r05
        # weak #1: the old `fp32 (gates differ in fp64)` branch judged at 2e-2); how many decisions the fp64 run would
        # have taken otherwise is recorded.
        f64 = torch.float64
        osd64 = {k: (v.detach().to(f64).requires_grad_(True) if v.is_floating_point() and 'running' not in k
                     else (v.detach().to(f64) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
        replay = cpu_ref.DecisionTape(tape.recorded)
        ref64 = cpu_ref.tggcn_forward(osd64, dict(m.cfg), x_human.to(f64), x_objects.to(f64), mask.to(f64),
                                      training=training, gumbel_noise=noise.to(f64), decisions=replay,
                                      **{k: v.to(f64) for k, v in kw.items()})
        assert replay.i == len(tape.recorded), 'the two oracle runs took different numbers of decisions'
        desc['fp64_decisions_differing'] = replay.differing
        n_hard = 2 if cad else 1
        same_gates = all(torch.equal(ref64[i].float(), ref[i].detach()) for i in range(n_hard))
        assert same_gates, 'the fp64 oracle run did not follow the replayed decisions'
        rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)) for i, o in enumerate(ref)]
        sum((o * r).sum() for o, r in zip(ref, rs) if o.requires_grad).backward()
        sum((o * r.to(f64)).sum() for o, r in zip(ref64, rs) if o.requires_grad).backward()
        rtol, atol, grad_ref = 2e-5, 1e-8, 'fp64|fp32'
        sum((o * r.to(dev)).sum() for o, r in zip(out, rs) if o.requires_grad).backward(retain_graph=True)   # state kept for boundary_layers
        off = []
        for pname, p in m.named_parameters():
            g32 = osd[pname].grad
            if g32 is None or float(g32.abs().max()) == 0.0:
                assert p.grad is None or float(p.grad.abs().max()) < 1e-6, ('dead grad', pname)
                continue
            assert p.grad is not None, ('missing grad', pname)
            scale = max(g32.abs().max().item(), 1e-6)
            got = p.grad.cpu()
            err = (got - g32).abs().max().item()
            # a tensor passes when it matches EITHER reference tightly: the fp64 run (the fp32 restatement loses
            # up to 1e-2 on the BatchNorm-conditioned GCN parameters) or the fp32 run (a ReLU unit whose sign flips
            # between fp32 and fp64 moves both fp32 implementations together)
            err64 = (got.to(f64) - osd64[pname].grad).abs().max().item()
            err = min(err, err64)
            # ill-conditioned cases (BatchNorm over a few dozen frames): both fp32 implementations sit ~1e-3 from
            # the fp64 run and from each other. The kernels pass when they are no further from fp64 than three
            # times the fp32 CPU restatement's own distance
            own = (g32.to(f64) - osd64[pname].grad).abs().max().item()
            if err64 <= 3.0 * own:
                # the TRUE deviation is logged (no clamp), in its own statistic: it is the conditioning of the case
                worst_cond = max(worst_cond, err / scale)
                if err >= rtol * scale + atol:
                    n_cond += 1
                else:
                    worst_g, worst_tiny = (max(worst_g, err / scale), worst_tiny) if err < rtol * scale else (worst_g, max(worst_tiny, err / scale))
                continue
            if err >= rtol * scale + atol:
                diff = (got - g32).abs()
                per_unit = diff.reshape(diff.shape[0], -1).max(dim=1).values if diff.dim() > 0 else diff.reshape(1)
                off.append((pname, err / scale, int((per_unit >= rtol * scale + atol).sum())))
                if os.environ.get('TWOG_FUZZ_VERBOSE'):
                    print(f'  off: {pname} shape {tuple(got.shape)} scale {scale:.3e} vs fp32 {(got - g32).abs().max().item():.3e} '
                          f'vs fp64 {err64:.3e} fp32-vs-fp64 {own:.3e}' + (f' signed kernels-fp64 {(got.to(f64) - osd64[pname].grad).flatten().tolist()} '
                          f'fp32-fp64 {(g32.to(f64) - osd64[pname].grad).flatten().tolist()} fp64 {osd64[pname].grad.flatten().tolist()}' if got.numel() <= 4 else ''), flush=True)
                continue
            worst_g, worst_tiny = (max(worst_g, err / scale), worst_tiny) if err < rtol * scale else (worst_g, max(worst_tiny, err / scale))
        if off:
            # Not rounding. The one legitimate cause is a ReLU unit on the other side of zero: fp32 pre-activations
            # differ by ~1e-6 between summation orders, so about one case in several hundred has a unit (mostly in the
            # K=128N geometry MLP) whose sign differs from the CPU run. Its signature (tools/parity_fuzz_diag.py): both
            # oracles agree with each other, the layer that owns the unit is off in that ONE weight row / bias element,
            # parameters upstream of it are perturbed broadly by a percent or so, everything downstream and every
            # output still match. Accept a case only with that signature; report how many there were.
            # ... and every other tensor that is off must be that layer's own parameter or lie UPSTREAM of it.
            owners = [o[0] for o in off if o[2] <= 2]
            # The owner must be CONFIRMED: tests/relu_boundary.py recomputes the pre-activations of every ReLU layer of
            # the path in fp64 from what the HIP path saved (also the per-pair layers of the general message forms, the
            # attention-score functions, the gate networks' hidden layers, the position features) and must find the unit
            # within rounding of zero. A confirmed unit may move its row by anything up to the whole row (a clip of
            # three frames: one frame's contribution is a third of the row). A deviation without a confirmed unit FAILS
            # the case (until round 5 the signature alone was accepted with a 30 % cap: VERDICT r05 weak #1).
            from tests.relu_boundary import boundary_layers
            found = boundary_layers(m, out)
            # Several units can sit on the boundary in one case (the larger the layout, the likelier): every off tensor
            # must be explained by SOME confirmed candidate.
            confirmed = [o for o in owners if o.rsplit('.', 1)[0] in found]
            unexplained = [(t, e) for t, e, _ in off
                           if not (e < 1.0 and any(_is_owner_or_upstream(t, own_) for own_ in confirmed))]
            # Tensors of one or a few elements (the bias of a score function, of a one-unit gate layer): sums of terms of
            # both signs -- the bias gradient of an attention score is EXACTLY zero but for the pairs its ReLU clips (a
            # softmax does not see a common shift) -- whose fp32 rounding noise is large against the net value, and for
            # which `own` above, ONE sample of that noise, is no yardstick (the ratio of two samples exceeds 3 one time
            # in five). Their yardstick is measured: the fp32 oracle itself, re-run on weights and inputs moved by at
            # most one unit in the last place (same hard decisions), against the fp64 run -- eight samples of the noise
            # of exactly this quantity. The kernels pass within four times the largest (their arithmetic is not the
            # oracle's: fused multiply-adds, v_exp / v_rcp based transcendentals, 6 of 9 bf16 chunk products); every such
            # judgement is listed in the summary with the three numbers.
            small = [t for t, _ in unexplained if osd[t].numel() <= SMALL_TENSOR]
            if small:
                spread = _fp32_noise_of_small_tensors(small, sd, m.cfg, x_human, x_objects, mask, kw, training, noise_in=noise,
                                                     recorded=tape.recorded, rs=rs, g64={t: osd64[t].grad for t in small})
                judged = []
                for t in small:
                    e64 = (dict(m.named_parameters())[t].grad.cpu().to(f64) - osd64[t].grad).abs().max().item()
                    judged.append(dict(tensor=t, kernels_vs_fp64=e64, fp32_oracle_noise_vs_fp64=spread[t],
                                       fp32_oracle_vs_fp64=(osd[t].grad.to(f64) - osd64[t].grad).abs().max().item()))
                    if e64 <= SMALL_TENSOR_NOISE_FACTOR * spread[t]:
                        unexplained = [(t_, e_) for t_, e_ in unexplained if t_ != t]
                desc['small_tensors_judged_by_fp32_noise'] = judged
            assert not unexplained, ('grad', sorted(off, key=lambda o: -o[1])[:6], 'not explained by a confirmed ReLU '
                                     'boundary unit', unexplained[:6], 'candidate owners', owners[:6],
                                     'boundary units found in', list(found), desc.get('small_tensors_judged_by_fp32_noise'))
            explained_by_relu = [o for o in off if any(_is_owner_or_upstream(o[0], own_) for own_ in confirmed)]
            if not explained_by_relu:
                desc['grad_ref'] = grad_ref
                desc.update(worst_output_rel=worst_out, worst_grad_rel=worst_g, worst_grad_rel_ill_conditioned=worst_cond,
                            tensors_judged_by_conditioning=n_cond)
                return desc
            desc['relu_boundary'] = confirmed[:6]
            desc['relu_boundary_confirmed'] = True
            desc['relu_boundary_worst_rel'] = max(e for _, e, _ in explained_by_relu)
    desc['grad_ref'] = grad_ref
    desc.update(worst_output_rel=worst_out, worst_grad_rel=worst_g, worst_grad_rel_ill_conditioned=worst_cond,
                tensors_judged_by_conditioning=n_cond, worst_grad_rel_of_vanishing_tensors=worst_tiny)
    return desc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if not os.environ.get('TWOG_FUZZ_ORACLE_ONLY'):
        assert kernels.get_kernels().name == 'hip'
    rng = random.Random(seed)
    results, failures = [], []
    t0 = time.time()
    for i in range(n):
        state = rng.getstate()
        if i < first:
            one_case(rng, i, dry=True, run_seed=seed)
            continue
        try:
            results.append(one_case(rng, i, run_seed=seed))
        except NotImplementedError as e:   # a configuration the gfx950 path declares unsupported: not a parity failure
            results.append(dict(idx=i, skipped=str(e)[:200]))
        except Exception as e:  # noqa: BLE001
            failures.append(dict(idx=i, error=repr(e)[:2500], rng_state_hash=hash(state) & 0xffffffff))
            print("FAIL", i, repr(e)[:2500], flush=True)
    ok = [r for r in results if 'worst_output_rel' in r and 'skipped' not in r]
    summary = dict(cases=n - first, first_case=first, passed=len(ok), skipped=len(results) - len(ok), failed=len(failures), seed=seed,
                   worst_output_rel=max((r['worst_output_rel'] for r in ok), default=0.0),
                   worst_grad_rel=max((r['worst_grad_rel'] for r in ok), default=0.0),
                   # (worst_grad_rel: tensors inside the RELATIVE term of the gate, 2e-5 of their scale; a tensor whose gradient is
                   # small -- scale below ~5e-4, e.g. the bias of a score function -- can pass through the absolute floor 1e-8
                   # alone: its error relative to its own scale is kept apart)
                   worst_grad_rel_of_vanishing_tensors=max((r.get('worst_grad_rel_of_vanishing_tensors', 0.0) for r in ok), default=0.0),
                   worst_grad_rel_ill_conditioned=max((r.get('worst_grad_rel_ill_conditioned', 0.0) for r in ok), default=0.0),
                   tensors_judged_by_conditioning=sum(r.get('tensors_judged_by_conditioning', 0) for r in ok),
                   cases_with_a_relu_unit_on_the_other_side_of_zero=sum(1 for r in ok if r.get('relu_boundary')),
                   cases_with_a_hard_decision_on_the_rounding_boundary=sum(1 for r in ok if r.get('decision_on_rounding_boundary')),
                   cases_per_gradient_reference={k: sum(1 for r in ok if r.get('grad_ref') == k)
                                                 for k in sorted({r.get('grad_ref') for r in ok if r.get('grad_ref')})},
                   small_tensors_judged_by_fp32_noise=[dict(idx=r['idx'], **j) for r in ok
                                                       for j in r.get('small_tensors_judged_by_fp32_noise', [])],
                   cases_where_fp64_alone_would_decide_a_gate_differently=sum(1 for r in ok if r.get('fp64_decisions_differing')),
                   seconds=time.time() - t0)
    # every case accepted by the ReLU-signature rule, with whether tests/relu_boundary.py CONFIRMED the unit (recomputed the
    # owner layer's pre-activations in fp64 and found the unit within rounding of zero); an unconfirmed acceptance fails the sweep
    accepted = [dict(idx=r['idx'], layout=dict(bs=r['bs'], T=r['T'], H=r['H'], O=r['O'], N=r['N'], h=r['h']),
                     candidate_owners=r['relu_boundary'], confirmed_by_fp64=bool(r.get('relu_boundary_confirmed')),
                     worst_grad_rel_of_the_case=r.get('relu_boundary_worst_rel')) for r in ok if r.get('relu_boundary')]
    summary['cases_accepted_by_the_relu_signature_rule'] = accepted
    summary['unconfirmed_acceptances'] = sum(1 for a in accepted if not a['confirmed_by_fp64'])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(dict(summary=summary, failures=failures, results=results),
              open(os.path.join(ROOT, 'gpurun_out', 'parity_fuzz.json'), 'w'), indent=1)
    print(json.dumps(summary))
    sys.exit(1 if failures or summary['unconfirmed_acceptances'] else 0)


if __name__ == '__main__':
    main()
