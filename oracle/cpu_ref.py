"""CPU oracle for the 2G-GCN hot path (TEST INFRASTRUCTURE -- never imported by the product package).

A functional restatement, in plain fp32 PyTorch-CPU ops, of ``TGGCN.forward`` of the reference
(/root/reference/vhoi/models.py:584-933) and of everything it calls. It keeps the reference's loop structure
(per time step, per entity, per sender), so that (a) equivalence with the reference is obvious line by line and
(b) its speed on host cores is representative of the reference's CPU path (bench.py ``cpu_baseline``, kind "port").

Parity status: PINNED. The reference has no tests or golden vectors of its own, so the oracle
is pinned against outputs of the reference itself: ``tools/make_golden.py`` imports /root/reference in the build
container, runs the real ``TGGCN`` forward+backward and ``Geo_gcn`` on deterministic inputs/weights
(``oracle/detgen.py``) and commits inputs + outputs + gradients under tests/golden/;
tests/test_oracle_golden.py checks this file against them.

The model is a dict ``sd`` of tensors with the reference's ``state_dict`` names and a dict ``cfg`` with the
reference's constructor keyword names (vhoi/models.py:179-190). Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg may use this module.
"""
import math
from collections import deque

import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(  # vhoi/models.py:179-190
    hidden_size=128, discrete_networks_num_layers=1, discrete_optimization_strategy='gumbel-sigmoid',
    filter_discrete_updates=False, gcn_node=26,
    message_humans_to_human=True, message_human_to_objects=True, message_objects_to_human=True,
    message_objects_to_object=True, message_geometry_to_objects=True, message_geometry_to_human=False,
    message_segment=False, message_type='relational', message_granularity='specific',
    message_aggregation='attention', attention_style='concat', object_segment_update_strategy='independent',
    update_segment_threshold=0.5, add_segment_length=False, add_time_position=False, time_position_strategy='s',
    positional_encoding_style='embedding', cat_level_states=False, share_level_mlps=False, bias=True)


def full_cfg(cfg: dict) -> dict:
    out = dict(DEFAULT_CFG)
    out.update(cfg)
    return out


# ----------------------------------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    """nn.Linear as built by build_mlp (pyrutils/torch/models.py:31-36)."""
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def _mlp_relu(sd, name, x, n_layers=1):
    """build_mlp([...], ['relu', ...]): Linear layers sit at Sequential indices 0, 2, 4 (no dropout)."""
    for i in range(n_layers):
        x = torch.relu(_lin(sd, f'{name}.{2 * i}', x))
    return x


def _gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell equations (gate order r, z, n)."""
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, dim=-1)
    h_r, h_z, h_n = gh.chunk(3, dim=-1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1.0 - z) * n + z * h


def _bigru(sd, name, x):
    """nn.GRU(h, h, 1 layer, batch_first, bidirectional) on (bs, T, h) with zero initial state
    (vhoi/models.py:998)."""
    bs, T, _ = x.shape
    hid = sd[name + '.weight_hh_l0'].shape[1]
    outs = []
    for suffix, order in (('', range(T)), ('_reverse', range(T - 1, -1, -1))):
        w_ih, w_hh = sd[f'{name}.weight_ih_l0{suffix}'], sd[f'{name}.weight_hh_l0{suffix}']
        b_ih, b_hh = sd.get(f'{name}.bias_ih_l0{suffix}'), sd.get(f'{name}.bias_hh_l0{suffix}')
        h = x.new_zeros(bs, hid)
        seq = [None] * T
        for t in order:
            h = _gru_cell(x[:, t], h, w_ih, w_hh, b_ih, b_hh)
            seq[t] = h
        outs.append(torch.stack(seq, dim=1))
    return torch.cat(outs, dim=-1)


# ----------------------------------------------------------------------------------------------------------------
# geometric-level GCN  (pyrutils/torch/models_gcn.py)
# ----------------------------------------------------------------------------------------------------------------
def geo_gcn(sd, x, training: bool, prefix='geometry_embedding_gcn', bn_state=None):
    """Geo_gcn.forward, models_gcn.py:30-37. x: (bs, 4, N, T) -> (bs, 128, N, T) contiguous.

    bn_state, if a dict, receives the updated BatchNorm running statistics (train mode), as torch would
    have written them in place (models_gcn.py:43-49).
    """
    bs, c, N, T = x.shape
    p = prefix + '.joint_embed.cnn'
    # norm_data: BatchNorm1d over channel = c*N + n   (models_gcn.py:45-50)
    xv = x.reshape(bs, c * N, T)
    gamma, beta = sd[p + '.0.bn.weight'], sd[p + '.0.bn.bias']
    if training:
        mean = xv.mean(dim=(0, 2))
        var_b = xv.var(dim=(0, 2), unbiased=False)
        if bn_state is not None:
            n = bs * T
            var_u = var_b * (n / max(n - 1, 1))
            with torch.no_grad():
                bn_state['running_mean'] = 0.9 * sd[p + '.0.bn.running_mean'] + 0.1 * mean
                bn_state['running_var'] = 0.9 * sd[p + '.0.bn.running_var'] + 0.1 * var_u
                bn_state['num_batches_tracked'] = sd[p + '.0.bn.num_batches_tracked'] + 1
    else:
        mean, var_b = sd[p + '.0.bn.running_mean'], sd[p + '.0.bn.running_var']
    xn = (xv - mean[None, :, None]) / torch.sqrt(var_b[None, :, None] + 1e-5)
    xn = xn * gamma[None, :, None] + beta[None, :, None]
    xn = xn.reshape(bs, c, N, T)
    # embed: two 1x1 convs == per-node Linear (models_gcn.py:57-63)
    w1, b1 = sd[p + '.1.cnn.weight'].flatten(1), sd[p + '.1.cnn.bias']
    w2, b2 = sd[p + '.3.cnn.weight'].flatten(1), sd[p + '.3.cnn.bias']
    e = xn.permute(0, 3, 2, 1)  # (bs, T, N, 4)
    e = torch.relu(F.linear(e, w1, b1))
    e = torch.relu(F.linear(e, w2, b2))  # (bs, T, N, 64)
    # compute_similarity (models_gcn.py:95-100): softmax over last dim, no 1/sqrt(d)
    ws1, bs1 = sd[prefix + '.get_s.s1.cnn.weight'].flatten(1), sd[prefix + '.get_s.s1.cnn.bias']
    ws2, bs2 = sd[prefix + '.get_s.s2.cnn.weight'].flatten(1), sd[prefix + '.get_s.s2.cnn.bias']
    q = F.linear(e, ws1, bs1)  # (bs, T, N, 128)
    k = F.linear(e, ws2, bs2)
    s = torch.softmax(q @ k.transpose(-1, -2), dim=-1)  # (bs, T, N, N)
    y = (s @ e) @ sd[prefix + '.weight']  # (bs, T, N, 128)
    return y.permute(0, 3, 2, 1).contiguous()  # (bs, 128, N, T)


def split_geometry(x_human):
    """vhoi/models.py:631-642 with the split generalised to 4N = F_h - 2048 (the reference hard-codes
    2124->76, 2168->120, else->104; all three satisfy the same formula). Geometry of human 0 only."""
    vw = x_human.shape[3] - 2048
    x_vis, x_geo = torch.split(x_human, [2048, vw], dim=-1)
    x_geo = x_geo[:, :, 0, :]
    bs, t, _ = x_geo.shape
    x_geo = x_geo.reshape(bs, t, vw // 4, 4).permute(0, 3, 2, 1).contiguous()
    return x_vis, x_geo


# ----------------------------------------------------------------------------------------------------------------
# messages / attention  (vhoi/models.py:1667-1775)
# ----------------------------------------------------------------------------------------------------------------
def relational_message(sd, receiver, senders, mask, f, g):
    rel = 0
    for s in range(senders.shape[1]):
        pair = torch.cat([receiver, senders[:, s]], dim=-1)
        rel = rel + _mlp_relu(sd, g, pair) * mask[:, s:s + 1]
    return _mlp_relu(sd, f, rel)


def non_relational_message(sd, receiver, senders, mask, granularity, fn):
    out = []
    for s in range(senders.shape[1]):
        snd = senders[:, s]
        if granularity in {'v2', 'specific'}:
            snd = torch.cat([receiver, snd], dim=-1)
        out.append(_mlp_relu(sd, fn, snd) * mask[:, s:s + 1])
    return torch.stack(out, dim=1)


def attention_weights(sd, query, keys, mask, style, fn=None):
    w = []
    for s in range(keys.shape[1]):
        key = keys[:, s]
        if style in {'v1', 'concat'}:
            a = _mlp_relu(sd, fn, torch.cat([query, key], dim=-1))
        elif style in {'v2', 'dot-product', 'v3', 'scaled_dot-product'}:
            a = torch.sum(query * key, dim=-1, keepdim=True)
            if style in {'v3', 'scaled_dot-product'}:
                a = a / math.sqrt(key.shape[-1])
        else:  # v4 / general: relu(Bilinear)
            a = torch.relu(F.bilinear(query, key, sd[fn + '.weight'], sd.get(fn + '.bias')))
        w.append(a)
    w = torch.cat(w, dim=-1)
    w = torch.where(mask.bool(), w, torch.full_like(w, float('-inf')))
    w = torch.softmax(w, dim=1)
    return torch.where(torch.isnan(w), torch.zeros_like(w), w)


def distance_attention_weights(dist, mask):
    dmask = dist.bool()
    ninf = torch.full_like(dist, float('-inf'))
    d = 1 / (dist + 1e-7)
    d = torch.where(mask.bool(), d, ninf)
    d = torch.where(dmask, d, ninf)
    w = torch.softmax(d, dim=-1)
    return torch.where(torch.isnan(w), torch.zeros_like(w), w)


def _message(sd, cfg, receiver, senders, mask, names, dists=None):
    """Common body of the ten *_message methods (vhoi/models.py:1004-1475).
    names = (relational f, relational g, message fn, attention fn). Returns (message, att weights or None)."""
    f, g, fn, att = names
    weights = None
    if cfg['message_type'] in {'v1', 'relational'}:
        m = relational_message(sd, receiver, senders, mask, f, g)
    else:
        m = non_relational_message(sd, receiver, senders, mask, cfg['message_granularity'], fn)
        if cfg['message_aggregation'] in {'mp', 'mean_pooling'}:
            n = torch.clamp(mask.sum(dim=1, keepdim=True), min=1.0)
            m = m.sum(dim=1) / n
        else:
            if dists is None:
                weights = attention_weights(sd, receiver, senders, mask, cfg['attention_style'], att)
            else:
                weights = distance_attention_weights(dists, mask)
            m = torch.sum(weights.unsqueeze(-1) * m, dim=1)
    return m, weights


def _drop(x, i, dim=1):
    idx = [j for j in range(x.shape[dim]) if j != i]
    return x.index_select(dim, torch.tensor(idx, dtype=torch.long)) if idx else x.narrow(dim, 0, 0)


_FRAME_NAMES = {  # relation -> (relational f, relational g, message fn, attention fn)   vhoi/models.py:323-520
    'hh': ('human_human_full_relation_mlp', 'human_human_pairwise_relation_mlp',
           'humans_to_human_message_mlp', 'humans_to_human_message_att_mlp'),
    'ho': ('object_human_full_relation_mlp', 'object_human_pairwise_relation_mlp',
           'human_to_object_message_mlp', 'humans_to_object_message_att_mlp'),
    'oh': ('human_object_full_relation_mlp', 'human_object_pairwise_relation_mlp',
           'objects_to_human_message_mlp', 'objects_to_human_message_att_mlp'),
    'oo': ('object_object_full_relation_mlp', 'object_object_pairwise_relation_mlp',
           'objects_to_object_message_mlp', 'objects_to_object_message_att_mlp'),
    'sh': ('human_geometry_full_relation_mlp', 'human_geometry_pairwise_relation_mlp',
           'geometry_to_human_message_mlp', 'geometry_to_human_message_att_mlp'),
    'so': ('object_geometry_full_relation_mlp', 'object_geometry_pairwise_relation_mlp',
           'geometry_to_object_message_mlp', 'geometry_to_object_message_att_mlp'),
}
_SEG_NAMES = {
    'hh': ('human_human_segment_full_relation_mlp', 'human_human_segment_pairwise_relation_mlp',
           'humans_to_human_segment_message_mlp', 'humans_to_human_segment_message_att_mlp'),
    'ho': ('object_human_segment_full_relation_mlp', 'object_human_segment_pairwise_relation_mlp',
           'human_to_object_segment_message_mlp', 'humans_to_object_segment_message_att_mlp'),
    'oh': ('human_object_segment_full_relation_mlp', 'human_object_segment_pairwise_relation_mlp',
           'objects_to_human_segment_message_mlp', 'objects_to_human_segment_message_att_mlp'),
    'oo': ('object_object_segment_full_relation_mlp', 'object_object_segment_pairwise_relation_mlp',
           'objects_to_object_segment_message_mlp', 'objects_to_object_segment_message_att_mlp'),
}


# ----------------------------------------------------------------------------------------------------------------
# discrete gates  (pyrutils/torch/distributions.py, vhoi/models.py:1620-1627)
# ----------------------------------------------------------------------------------------------------------------
class GumbelSource:
    """Hands out Gumbel(0,1) noise of shape (bs, 2) per call, in the reference's call order (t-major, humans then
    objects; distributions.py:16). With ``noise`` given (tensor (calls, bs, 2)) it replays it, otherwise it draws from
    the CPU default generator exactly as the reference does, and records what it drew in ``drawn``."""

    def __init__(self, noise=None):
        self.noise, self.i, self.drawn = noise, 0, []

    def __call__(self, size):
        if self.noise is not None:
            g = self.noise[self.i]
            self.i += 1
            assert tuple(g.shape) == tuple(size)
        else:
            g = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample(size)
        self.drawn.append(g)
        return g


class DecisionTape:
    """Test aid for comparing two runs of this file at different precisions (tools/parity_fuzz.py: fp32 and fp64). Every
    HARD decision of the path -- `soft > threshold` of the estimators, the three comparisons of the local-maximum filter
    -- passes through the tape: a recording tape keeps the boolean tensors of a run in call order, a replaying tape hands
    the recorded ones back in place of the second run's own comparisons, so that both runs follow the same discrete
    path (and `differing` counts the elements the second run would have decided otherwise). Without a tape (the default)
    every comparison is used as computed -- the behaviour the golden vectors pin."""

    def __init__(self, recorded=None):
        self.replay = recorded is not None
        self.recorded = [] if recorded is None else list(recorded)
        self.i, self.differing = 0, 0

    def __call__(self, cond):
        if not self.replay:
            self.recorded.append(cond.detach().clone())
            return cond
        got = self.recorded[self.i]
        self.i += 1
        assert got.shape == cond.shape, (got.shape, cond.shape)
        self.differing += int((got != cond).sum())
        return got


def _as_computed(cond):
    return cond


def discrete_estimator(p, strategy, threshold, gumbel, decide=_as_computed):
    if strategy in {'straight-through', 'st'}:
        z = decide(p > threshold).to(p.dtype)
        return (z - p).detach() + p, p  # forward value z, d/dp = 1 (distributions.py:39-53)
    elif strategy in {'gumbel-sigmoid', 'gs'}:
        pp = torch.cat([p, 1.0 - p], dim=-1)
        y = torch.log(pp + 1e-20) + gumbel(pp.size())
        y = torch.softmax(y / 1.0, dim=-1)[:, :1]
        z = decide(y > threshold).to(y.dtype)
        return (z - y).detach() + y, y
    raise ValueError(f'strategy must be either straight-through or gumbel-sigmoid, not {strategy}.')


def _gate_mlp(sd, name, x, n_layers):
    for i in range(n_layers - 1):
        x = torch.relu(_lin(sd, f'{name}.{2 * i}', x))
    return torch.sigmoid(_lin(sd, f'{name}.{2 * (n_layers - 1)}', x))


def filter_soft_decisions(ux_s, thr, decide=_as_computed):
    """vhoi/models.py:1637-1664."""
    out = []
    T = len(ux_s)
    for t in range(T):
        u = ux_s[t]
        um1 = ux_s[t - 1] if t else torch.zeros_like(u)
        up1 = ux_s[t + 1] if t + 1 < T else torch.zeros_like(u)
        cond = decide(u > um1) & decide(u > up1) & decide(u >= thr)
        uh = decide(u >= thr).to(u.dtype)
        uh = (uh - u).detach() + u
        out.append(torch.where(cond, uh, torch.clamp(uh, max=0.0)))
    return out


def reorder_hidden_states(hx, ux):
    """vhoi/models.py:1567-1586. hx (bs, T, d), ux (bs, T): frames strictly inside a segment take the state of the
    segment's end frame; frames after the last end flag keep their own state."""
    bs, T, _ = hx.shape
    rows = []
    for m in range(bs):
        ends = [-1] + torch.nonzero(ux[m], as_tuple=True)[0].tolist()
        idx = list(range(T))
        for s, e in zip(ends[:-1], ends[1:]):
            for t in range(s + 1, e):
                idx[t] = e
        rows.append(hx[m, idx])
    return torch.stack(rows, dim=0)


def time_tensor(steps_per_example, T, ignore_division=False):
    x = torch.arange(1, T + 1, dtype=steps_per_example.dtype).unsqueeze(-1)
    x = torch.repeat_interleave(x, repeats=steps_per_example.shape[0], dim=1)
    if not ignore_division:
        x = x / steps_per_example
    return x.unsqueeze(-1)  # (T, bs, 1)


def segment_length_tensor(ux_s, steps_per_example, ignore_division=False):
    T, bs = len(ux_s[0]), ux_s[0][0].shape[0]
    x_time = time_tensor(steps_per_example, T, ignore_division)
    out = []
    for ux_se in ux_s:
        acc = torch.zeros(bs, 1, dtype=ux_s[0][0].dtype)
        rows = []
        for u, xt in zip(ux_se, x_time):
            rel = u * xt
            rel = torch.where(rel.bool(), rel - acc, rel)
            acc = acc + rel
            rows.append(rel)
        out.append(torch.cat(rows, dim=-1))
    return torch.stack(out, dim=-1).unsqueeze(-1)  # (bs, T, E, 1)


def periodic_embedding(x, hidden):
    w = torch.tensor([1e4], dtype=x.dtype) ** torch.linspace(0, 1, hidden // 2, dtype=x.dtype)
    return torch.cat([torch.sin(x / w), torch.cos(x / w)], dim=-1)


# ----------------------------------------------------------------------------------------------------------------
# TGGCN.forward  (vhoi/models.py:584-933)
# ----------------------------------------------------------------------------------------------------------------
def tggcn_forward(sd, cfg, x_human, x_objects, objects_mask, human_segmentation=None, objects_segmentation=None,
                  human_human_distances=None, human_object_distances=None, object_object_distances=None,
                  steps_per_example=None, inspect_model=False, training=True, gumbel_noise=None, aux=None,
                  decisions=None):
    """Returns the reference's output list (6 tensors without affordance heads, 12 with). ``aux`` (a dict) receives
    intermediates used by the unit tests: bn_state, gumbel noise drawn, geometry feature, xx_hs/xx_os ...
    ``decisions``: an optional DecisionTape (see there); None = every comparison as computed."""
    decide = _as_computed if decisions is None else decisions
    cfg = full_cfg(cfg)
    aux = {} if aux is None else aux
    hid = cfg['hidden_size']
    gumbel = GumbelSource(gumbel_noise)
    H, O = x_human.shape[2], x_objects.shape[2]
    has_aff = 'object_recognition_mlp.0.weight' in sd
    n_gate = cfg['discrete_networks_num_layers']
    strat, thr = cfg['discrete_optimization_strategy'], cfg['update_segment_threshold']

    # A. geometry preamble + GCN (models.py:631-645); the .view at :645 is a raw reinterpretation (Appendix A1).
    x_vis, x_geo = split_geometry(x_human)
    bs, T = x_vis.shape[0], x_vis.shape[1]
    bn_state = {}
    g = geo_gcn(sd, x_geo, training, bn_state=bn_state)
    aux['bn_state'], aux['gcn_out'] = bn_state, g
    x_geometry = g.reshape(bs, T, 1, g.shape[1] * g.shape[2])  # flat (c, n, t) block re-read as T rows of 128N
    # B. embeddings (models.py:646)
    x_h = _mlp_relu(sd, 'human_embedding_mlp', x_vis)
    x_o = _mlp_relu(sd, 'object_embedding_mlp', x_objects)
    x_s = _mlp_relu(sd, 'geometry_embedding_mlp', x_geometry, n_layers=2)

    # C. frame-level BiGRUs (models.py:983-1002)
    def frame_rnn(x, rnn, emb):
        h_fr = torch.stack([_bigru(sd, rnn, x[:, :, e]) for e in range(x.shape[2])], dim=2)
        return _mlp_relu(sd, emb, h_fr), h_fr

    h_hf, h_hfr = frame_rnn(x_h, 'human_bd_rnn', 'human_bd_embedding_mlp')
    h_of, h_ofr = frame_rnn(x_o, 'object_bd_rnn', 'object_bd_embedding_mlp')
    h_sf, h_sfr = frame_rnn(x_s, 'geometry_bd_rnn', 'geometry_bd_embedding_mlp')
    aux.update(x_h=x_h, x_o=x_o, x_s=x_s, h_hf=h_hf, h_of=h_of, h_sf=h_sf, h_hfr=h_hfr, h_ofr=h_ofr)

    periodic = cfg['positional_encoding_style'] in {'p', 'periodic'}

    def pos_embed(x, mlp):
        return periodic_embedding(x, hid) if periodic else _mlp_relu(sd, mlp, x)

    x_time_u = None
    if cfg['add_time_position'] and cfg['time_position_strategy'] == 'u':
        x_time_u = pos_embed(time_tensor(steps_per_example, T, periodic), 'time_position_mlp').transpose(0, 1)

    # D. frame-level loop (models.py:664-749)
    xx_hs, xx_os = [[] for _ in range(H)], [[] for _ in range(O)]
    ux_hs, ux_os = [[] for _ in range(H)], [[] for _ in range(O)]
    ux_hss, ux_oss = [[] for _ in range(H)], [[] for _ in range(O)]
    ax_hf = [[] for _ in range(H)]
    ones_h = torch.ones(bs, H, dtype=x_h.dtype)
    for t in range(T):
        x_tt = x_time_u[:, t] if x_time_u is not None else None
        f_h = torch.cat([x_h[:, t], h_hf[:, t]], dim=-1)  # (bs, H, 2h)
        f_o = torch.cat([x_o[:, t], h_of[:, t]], dim=-1)
        f_s = torch.cat([x_s[:, t], h_sf[:, t]], dim=-1)  # (bs, 1, 2h)
        for h in range(H):
            parts = [h_hf[:, t, h]]
            gate_in = [x_h[:, t, h], h_hf[:, t, h]]
            if cfg['message_humans_to_human']:
                d = None
                if human_human_distances is not None:
                    d = _drop(human_human_distances[:, t, h], h, dim=-1)
                m, _ = _message(sd, cfg, f_h[:, h], _drop(f_h, h), _drop(ones_h, h), _FRAME_NAMES['hh'], d)
                parts.append(m)
                gate_in.append(m)
            if cfg['message_objects_to_human']:
                d = human_object_distances[:, t, h] if human_object_distances is not None else None
                m, w = _message(sd, cfg, f_h[:, h], f_o, objects_mask, _FRAME_NAMES['oh'], d)
                ax_hf[h].append(w)
                parts.append(m)
                gate_in.append(m)
            if cfg['message_geometry_to_human']:
                m, _ = _message(sd, cfg, f_h[:, h], f_s, torch.ones(bs, 1), _FRAME_NAMES['sh'])
                parts.append(m)
                gate_in.append(m)
            if human_segmentation is not None:
                u = us = human_segmentation[:, t:t + 1, h]
            else:
                gi = gate_in + ([x_tt] if x_tt is not None else [])
                p = _gate_mlp(sd, 'update_human_segment_mlp', torch.cat(gi, dim=-1), n_gate)
                u, us = discrete_estimator(p, strat, thr, gumbel, decide)
                if t == T - 1:
                    u = torch.ones_like(u)  # in-place override, cuts the gradient (models.py:701-702)
            ux_hs[h].append(u)
            ux_hss[h].append(us)
            xx_hs[h].append(torch.cat(parts, dim=-1))
        for k in range(O):
            parts = [h_of[:, t, k]]
            m_ho = m_so = m_oo = None
            if cfg['message_human_to_objects']:
                d = human_object_distances[:, t, :, k] if human_object_distances is not None else None
                m_ho, _ = _message(sd, cfg, f_o[:, k], f_h, ones_h, _FRAME_NAMES['ho'], d)
                m_ho = m_ho * objects_mask[:, k:k + 1]
                parts.append(m_ho)
            if cfg['message_geometry_to_objects']:
                m_so, _ = _message(sd, cfg, f_o[:, k], f_s, torch.ones(bs, 1), _FRAME_NAMES['so'])
                m_so = m_so * objects_mask[:, k:k + 1]
                parts.append(m_so)
            if cfg['message_objects_to_object']:
                d = None
                if object_object_distances is not None:
                    d = _drop(object_object_distances[:, t, k], k, dim=-1)
                m_oo, _ = _message(sd, cfg, f_o[:, k], _drop(f_o, k), _drop(objects_mask, k), _FRAME_NAMES['oo'], d)
                parts.append(m_oo)
            if objects_segmentation is not None:
                u = us = objects_segmentation[:, t:t + 1, k]
            else:
                u_h = ux_hs[0][-1] if H == 1 else None
                u_hs = ux_hss[0][-1] if H == 1 else None
                ostrat = cfg['object_segment_update_strategy']
                if ostrat in {'same_as_human', 'sah'} and u_h is not None and u_hs is not None:
                    u, us = u_h, u_hs
                else:  # gate input order differs from xx_os order: [x, h, m_ho, m_oo, m_so]  (models.py:1527)
                    gi = [t_ for t_ in [x_o[:, t, k], h_of[:, t, k], m_ho, m_oo, m_so, x_tt] if t_ is not None]
                    p = _gate_mlp(sd, 'update_object_segment_mlp', torch.cat(gi, dim=-1), n_gate)
                    u, us = discrete_estimator(p, strat, thr, gumbel, decide)
                    if ostrat in {'conditional_on_human', 'coh'} and u_h is not None:
                        u = u * u_h
                if t == T - 1:
                    u = torch.ones_like(u)
            ux_os[k].append(u)
            ux_oss[k].append(us)
            xx_os[k].append(torch.cat(parts, dim=-1))
    # E. optional filters / position features (models.py:751-779)
    if cfg['filter_discrete_updates']:
        ux_hs = [filter_soft_decisions(u, thr, decide) for u in ux_hss]
        ux_os = [filter_soft_decisions(u, thr, decide) for u in ux_oss]
    if cfg['add_time_position'] and cfg['time_position_strategy'] == 's':
        x_time = pos_embed(time_tensor(steps_per_example, T, periodic), 'time_position_mlp')
        xx_hs = [[torch.cat([a, b], dim=-1) for a, b in zip(xs, x_time)] for xs in xx_hs]
        xx_os = [[torch.cat([a, b], dim=-1) for a, b in zip(xs, x_time)] for xs in xx_os]
    if cfg['add_segment_length']:
        for xx, ux in ((xx_hs, ux_hs), (xx_os, ux_os)):
            if not xx:
                continue
            sl = pos_embed(segment_length_tensor(ux, steps_per_example, periodic), 'segment_length_mlp')
            sl = sl.permute(2, 1, 0, 3)
            for e in range(len(xx)):
                xx[e] = [torch.cat([a, b], dim=-1) for a, b in zip(xx[e], sl[e])]
    aux.update(xx_hs=xx_hs, xx_os=xx_os)

    # F. segment-level loop (models.py:785-880)
    cells = {('h', 'f'): 'human_segment_rnn_fcell', ('h', 'b'): 'human_segment_rnn_bcell',
             ('o', 'f'): 'object_segment_rnn_fcell', ('o', 'b'): 'object_segment_rnn_bcell'}

    def cell(kind, direction, x, u, h_prev):
        n = cells[(kind, direction)]
        new = _gru_cell(x, h_prev, sd[n + '.weight_ih'], sd[n + '.weight_hh'], sd.get(n + '.bias_ih'),
                        sd.get(n + '.bias_hh'))
        return u * new + (1.0 - u) * h_prev  # models.py:1555

    zeros = torch.zeros(bs, hid, dtype=x_h.dtype)
    hx_hsf, hx_hsb = [[] for _ in range(H)], [deque() for _ in range(H)]
    hx_osf, hx_osb = [[] for _ in range(O)], [deque() for _ in range(O)]
    ax_hsf, ax_hsb = [[] for _ in range(H)], [deque() for _ in range(H)]

    def last(hx, direction):
        if len(hx) == 0:
            return zeros
        return hx[-1] if direction == 'f' else hx[0]

    for tf in range(T):
        tb = T - 1 - tf
        new_h = {'f': [], 'b': []}
        new_o = {'f': [], 'b': []}
        for direction, tt, hx_h, hx_o, ax in (('f', tf, hx_hsf, hx_osf, ax_hsf), ('b', tb, hx_hsb, hx_osb, ax_hsb)):
            prev_h = [last(hx_h[h], direction) for h in range(H)]
            prev_o = [last(hx_o[k], direction) for k in range(O)]
            sh = torch.stack(prev_h, dim=1) if H else None
            so = torch.stack(prev_o, dim=1) if O else None
            for h in range(H):
                x = xx_hs[h][tt]
                if cfg['message_segment']:
                    if cfg['message_humans_to_human']:
                        d = None
                        if human_human_distances is not None:
                            d = _drop(human_human_distances[:, tt, h], h, dim=-1)
                        m, _ = _message(sd, cfg, prev_h[h], _drop(sh, h), _drop(ones_h, h), _SEG_NAMES['hh'], d)
                        x = torch.cat([x, m], dim=-1)
                    if cfg['message_objects_to_human']:
                        d = human_object_distances[:, tt, h] if human_object_distances is not None else None
                        m, w = _message(sd, cfg, prev_h[h], so, objects_mask, _SEG_NAMES['oh'], d)
                        if direction == 'f':
                            ax[h].append(w)
                        else:
                            ax[h].appendleft(w)
                        x = torch.cat([x, m], dim=-1)
                new_h[direction].append(cell('h', direction, x, ux_hs[h][tt], prev_h[h]))
            for k in range(O):
                x = xx_os[k][tt]
                if cfg['message_segment']:
                    if cfg['message_human_to_objects']:
                        d = human_object_distances[:, tt, :, k] if human_object_distances is not None else None
                        m, _ = _message(sd, cfg, prev_o[k], sh, ones_h, _SEG_NAMES['ho'], d)
                        x = torch.cat([x, m], dim=-1)
                    if cfg['message_objects_to_object']:
                        d = None
                        if object_object_distances is not None:
                            d = _drop(object_object_distances[:, tt, k], k, dim=-1)
                        m, _ = _message(sd, cfg, prev_o[k], _drop(so, k), _drop(objects_mask, k), _SEG_NAMES['oo'], d)
                        x = torch.cat([x, m], dim=-1)
                new_o[direction].append(cell('o', direction, x, ux_os[k][tt], prev_o[k]))
        for h in range(H):  # commit (models.py:874-880)
            hx_hsf[h].append(new_h['f'][h])
            hx_hsb[h].appendleft(new_h['b'][h])
        for k in range(O):
            hx_osf[k].append(new_o['f'][k])
            hx_osb[k].appendleft(new_o['b'][k])

    # G. cat fwd|bwd, reorder (models.py:881-899)
    def finish(hx_f, hx_b, ux):
        per_e = []
        for e in range(len(hx_f)):
            hx = torch.stack([torch.cat([a, b], dim=-1) for a, b in zip(hx_f[e], hx_b[e])], dim=1)
            u = torch.cat(ux[e], dim=-1)
            per_e.append(reorder_hidden_states(hx, u.detach()))
        return torch.stack(per_e, dim=2)

    hx_hs = finish(hx_hsf, hx_hsb, ux_hs)
    hx_os = finish(hx_osf, hx_osb, ux_os) if O else None
    aux.update(hx_hs=hx_hs, hx_os=hx_os)
    if cfg['cat_level_states']:
        hx_hs = torch.cat([hx_hs, h_hfr], dim=-1)
        hx_os = torch.cat([hx_os, h_ofr], dim=-1)

    # H. heads (models.py:905-926)
    def head(name, x):
        return torch.log_softmax(_lin(sd, name + '.0', x), dim=-1).permute(0, 3, 1, 2).contiguous()

    def frame_head(kind, x):
        if cfg['share_level_mlps'] and not cfg['cat_level_states']:
            return head(kind.replace('_frame', ''), x)
        return head(kind, x)

    y_hs = torch.stack([torch.cat(u, dim=-1) for u in ux_hs], dim=-1)
    y_hss = torch.stack([torch.cat(u, dim=-1) for u in ux_hss], dim=-1)
    out_h = [frame_head('human_frame_recognition_mlp', h_hfr), frame_head('human_frame_prediction_mlp', h_hfr),
             head('human_recognition_mlp', hx_hs), head('human_prediction_mlp', hx_hs)]
    if not has_aff:
        output = [y_hs, y_hss] + out_h
    else:
        y_os = torch.stack([torch.cat(u, dim=-1) for u in ux_os], dim=-1)
        y_oss = torch.stack([torch.cat(u, dim=-1) for u in ux_oss], dim=-1)
        out_o = [frame_head('object_frame_recognition_mlp', h_ofr), frame_head('object_frame_prediction_mlp', h_ofr),
                 head('object_recognition_mlp', hx_os), head('object_prediction_mlp', hx_os)]
        output = [y_hs, y_os, y_hss, y_oss, out_h[0], out_h[1], out_o[0], out_o[1],
                  out_h[2], out_h[3], out_o[2], out_o[3]]
    aux['gumbel_drawn'] = gumbel.drawn
    aux['ux_os'] = ux_os
    aux['ux_oss'] = ux_oss
    if inspect_model:
        a_f = torch.stack([torch.stack(a, dim=1) for a in ax_hf], dim=1)
        a_sf = torch.stack([torch.stack(a, dim=1) for a in ax_hsf], dim=1)
        a_sb = torch.stack([torch.stack(list(a), dim=1) for a in ax_hsb], dim=1)
        return output, [a_f, a_sf, a_sb]
    return output


# ----------------------------------------------------------------------------------------------------------------
# losses used to form the fwd+bwd scalar (pyrutils/torch/losses.py:7-51, vhoi/losses.py:8-62)
# ----------------------------------------------------------------------------------------------------------------
def bce_loss(inp, target, ignore_value=-1):
    mask = (target != ignore_value).float()
    n = mask.sum().item()
    if n == 0:
        return torch.tensor(0.0, dtype=inp.dtype)
    return F.binary_cross_entropy(inp * mask, target * mask) * (inp.numel() / n)


def budget_loss(inp, target, ignore_value=-1):
    mask = (target != ignore_value).float()
    n = mask.sum().item()
    if n == 0:
        return torch.tensor(0.0, dtype=inp.dtype)
    return torch.mean(inp * mask) * (inp.numel() / n)


def loss_list(outputs, targets, weights, cad120: bool):
    """multi_task_loss with the 2G-GCN loss layout: non-CAD [B_HS, BCE_HS, NLL x4]; CAD [B x2, BCE x2, NLL x8]."""
    n_b = 2 if cad120 else 1
    fns = [budget_loss] * n_b + [bce_loss] * n_b + [None] * (8 if cad120 else 4)
    out = []
    for o, t, fn, w in zip(outputs, targets, fns, weights):
        if fn is None:
            out.append(w * F.nll_loss(o, t, ignore_index=-1))
        else:
            out.append(w * fn(o, t))
    return out
