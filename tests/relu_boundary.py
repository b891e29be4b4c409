"""Test infrastructure: which ReLU layers of a forward pass have units sitting ON the boundary, i.e. a pre-activation
whose magnitude is below the fp32 rounding error of its own dot product, so that the summation order alone decides on
which side of zero it lands (CPU oracle and HIP kernels sum in different orders).

A gradient comparison may tolerate a deviation only where this module finds such a unit: the owning layer's gradient
moves in that unit's row, the gradients UPSTREAM of the layer are perturbed, everything downstream and every output
stays put (tools/parity_fuzz.py::_is_owner_or_upstream). The pre-activations are recomputed in fp64 (torch on the
device is the checker here) from the activations the HIP path itself saved for its backward pass."""
import torch

U = 2.0 ** -24   # fp32 unit round-off


def _near(pre, mag, width, with_band=False):
    """pre, mag (rows, N) fp64: pre-activations and the sum of the magnitudes of their terms -> per output unit, the number
    of rows whose pre-activation is within `width` * u * mag of zero (a few units of the typical error of a K-term fp32
    dot product, far below the worst-case K u bound). with_band: also the widest such band per unit (what a bias nudge
    has to clear). Rows with mag == 0 (pairs that do not exist: a masked sender, the receiver itself) never count."""
    band = width * U * mag
    near = (pre.abs() <= band) & (mag > 0)
    if with_band:
        return near.sum(0), torch.where(near, band, torch.zeros_like(band)).max(0).values
    return near.sum(0)   # (N,)


def _boundary_units(x, w, b, width=4.0, with_band=False):
    """x (rows, K), w (N, K), b (N,) or None: _near of the layer's pre-activations, recomputed in fp64."""
    return _near(*_Rows(None, x, w, b).pre_mag(), width, with_band)


class _Rows:
    """A ReLU layer applied to stored rows: relu(x W^T + b)."""

    def __init__(self, layer, x, w, b, factor=1.0):
        self.layer, self.factor, self.units = layer, factor, int(w.shape[0])
        self.x, self.w, self.b = x.reshape(-1, x.shape[-1]), w.reshape(w.shape[0], -1), b
        self.rows = int(self.x.shape[0])

    def pre_mag(self):
        x, w = self.x.double(), self.w.double()
        pre, mag = x @ w.t(), x.abs() @ w.abs().t()
        if self.b is not None:
            pre, mag = pre + self.b.double(), mag + self.b.double().abs()
        return pre, mag


class _Pairs:
    """A ReLU layer applied to cat[receiver, sender] of every (receiver, sender) pair of an instance (a frame, or one
    step of one direction of the segment level): relu(W [f_r ; f_s] + b) -- the receiver-specific message function
    (vhoi/models.py:1712-1713), the pairwise relation g (:1683-1685) and the concat attention score (:1735-1737). The HIP
    path applies the two halves of W separately and adds them per pair inside relation.hip; here the halves are formed in
    fp64 and added by broadcasting. fr (I, R, D), fs (I, S, D), valid (I, R, S) bool: pairs that exist (not the receiver
    itself, not a masked sender -- their result is multiplied by 0 / replaced by -inf whatever the sign)."""

    def __init__(self, layer, fr, fs, valid, w, b, factor=1.0):
        self.layer, self.factor, self.units = layer, factor, int(w.shape[0])
        self.fr, self.fs, self.valid, self.w, self.b = fr, fs, valid, w.reshape(w.shape[0], -1), b
        self.rows = int(valid.sum())

    def pre_mag(self):
        D = self.fr.shape[-1]
        fr, fs, w = self.fr.double(), self.fs.double(), self.w.double()
        wr, ws = w[:, :D], w[:, D:]
        pre_s, mag_s = fs @ ws.t(), fs.abs() @ ws.abs().t()
        if self.b is not None:
            pre_s, mag_s = pre_s + self.b.double(), mag_s + self.b.double().abs()
        pre = (fr @ wr.t()).unsqueeze(2) + pre_s.unsqueeze(1)            # (I, R, S, N)
        mag = (fr.abs() @ wr.abs().t()).unsqueeze(2) + mag_s.unsqueeze(1)
        mag = mag * self.valid.unsqueeze(-1)
        return pre.reshape(-1, self.units), mag.reshape(-1, self.units)


class _Bilinear:
    """relu(q^T A k + b) of every (receiver, sender) pair: the `general` attention score (vhoi/models.py:1746; nn.Bilinear
    with one output). One unit; the magnitude of its terms is |q|^T |A| |k| + |b|."""

    def __init__(self, layer, fr, fs, valid, w, b):
        self.layer, self.factor, self.units = layer, 1.0, 1
        self.fr, self.fs, self.valid, self.w, self.b = fr, fs, valid, w.reshape(w.shape[-2], w.shape[-1]), b
        self.rows = int(valid.sum())

    def pre_mag(self):
        fr, fs, w = self.fr.double(), self.fs.double(), self.w.double()
        pre = torch.einsum('ird,isd->irs', fr @ w, fs)
        mag = torch.einsum('ird,isd->irs', fr.abs() @ w.abs(), fs.abs())
        if self.b is not None:
            pre, mag = pre + self.b.double(), mag + self.b.double().abs()
        mag = mag * self.valid
        return pre.reshape(-1, 1), mag.reshape(-1, 1)


# the first 1x1 convolution of the geometric-level GCN reads the BatchNorm output: the two implementations' batch
# statistics differ by rounding (fp64 ordered sums vs torch's), which moves its pre-activations by more than the
# rounding of its own 4-term dot product -- its band is this many times wider
GCN_CONV1_WIDTH_FACTOR = 16.0


def _relation_layers(plan, P, ops, rel, segment, fr, fs, mask_s, agg):
    """The ReLU layers of one relation at one level, in whichever form the configuration gives the messages
    (vhoi/models.py:1025-1049 and the nine sibling methods). fr (I, R, D) receivers, fs (I, S, D) senders, mask_s (I, S)
    the senders' object mask or None, agg: the summed pairwise relations (I*R, h) (relational messages only)."""
    I, R, S = fr.shape[0], fr.shape[1], fs.shape[1]
    if I == 0 or R == 0 or S == 0:
        return []
    rk, sk = ops._REL_ENDS[rel]
    seg = '_segment' if segment else ''
    msg = (ops._SEG_MLP if segment else ops._FRAME_MLP)[rel] + '.0'
    att = ops._ATT_MLP[rel].replace('_message_att_mlp', seg + '_message_att_mlp')
    valid = torch.ones(I, R, S, dtype=torch.bool, device=fr.device)
    if rel in ('hh', 'oo'):
        valid &= ~torch.eye(R, dtype=torch.bool, device=fr.device).unsqueeze(0)
    if mask_s is not None:
        valid &= (mask_s != 0).unsqueeze(1)
    out = []

    def par(name):
        return P.get(name + '.weight'), P.get(name + '.bias')

    if plan.relational:
        g, f = (f'{ops._REL_PREFIX[rel]}{seg}_{kind}_relation_mlp.0' for kind in ('pairwise', 'full'))
        out.append(_Pairs(g, fr, fs, valid, *par(g)))
        if agg is not None:
            out.append(_Rows(f, agg, *par(f)))
        return out
    if plan.specific:
        out.append(_Pairs(msg, fr, fs, valid, *par(msg)))
    else:
        out.append(_Rows(msg, fs, *par(msg)))
    d = plan.dists or {}
    by_distance = d.get({'hh': 'hh', 'oh': 'ho', 'ho': 'ho', 'oo': 'oo'}.get(rel)) is not None
    if sk == 's' or plan.mean_pool or by_distance or plan.att_style == 'dot':
        return out   # one sender (weight 1 whatever the score), mean pooling, distances, dot products: no score function
    if plan.att_style == 'concat':
        out.append(_Pairs(att + '.0', fr, fs, valid, *par(att + '.0')))
    else:
        out.append(_Bilinear(att, fr, fs, valid, *par(att)))
    return out


def _layers(model, out):
    """Every ReLU layer of the forward pass, as objects with .layer (the name in the state_dict, without '.weight'),
    .units, .rows, .factor (band width factor) and .pre_mag() -> fp64 (rows, units) pre-activations and term magnitudes,
    recomputed from what the HIP path saved for its backward pass (where it did not save the operand -- the second GCN
    convolution, the per-pair sums of the general message forms -- from the operand's own inputs)."""
    from twog_gcn_amd import ops
    node = next(o.grad_fn for o in out if o.grad_fn is not None)
    S, plan, P = ops.saved_state(node), node.plan, dict(model.named_parameters())
    x_human, x_objects, objects_mask = node.inputs
    bs, T, H, O, N, h = plan.bs, plan.T, plan.H, plan.O, plan.N, plan.h
    nF = bs * T
    HUMv, OBJv, GEOv = (S[k].view(-1, S[k].shape[-1]) for k in ('HUM', 'OBJ', 'GEO'))
    items = []

    def rows(layer, x, factor=1.0):
        w = P.get(layer + '.weight')
        if w is None or x is None:
            return   # a configuration without this layer
        assert x.shape[-1] == w[0].numel(), ('operand layout of a ReLU layer not covered', layer, x.shape, w.shape)
        items.append(_Rows(layer, x, w, P.get(layer + '.bias'), factor))

    rows('human_embedding_mlp.0', x_human.view(nF * H, -1)[:, :2048])
    if O:
        rows('object_embedding_mlp.0', x_objects.view(nF * O, -1))
    rows('geometry_embedding_mlp.0', S['Gout'].view(nF, 128 * N))
    rows('geometry_embedding_mlp.2', S.get('t1'))
    if S.get('ab') is not None:
        # x^[f, n, c] = a[c*N + n] * x[f, n, c] + b[c*N + n] (BatchNorm folded, models_gcn.py:45-50), geometry of human 0
        ab = S['ab'].view(2, 4, N).double()
        xg = x_human.view(nF, H, -1)[:, 0, 2048:].reshape(nF, N, 4).double()
        xhat = xg * ab[0].t().unsqueeze(0) + ab[1].t().unsqueeze(0)
        rows('geometry_embedding_gcn.joint_embed.cnn.1.cnn', xhat.reshape(nF * N, 4), GCN_CONV1_WIDTH_FACTOR)
        # the second convolution's input e1 = relu(W1 x^ + b1) is not stored by the fused forward kernel: recomputed here
        c1 = 'geometry_embedding_gcn.joint_embed.cnn.1.cnn'
        w1, b1 = P[c1 + '.weight'].view(64, 4).double(), P[c1 + '.bias'].double()
        e1 = torch.relu(xhat.reshape(nF * N, 4) @ w1.t() + b1)
        rows('geometry_embedding_gcn.joint_embed.cnn.3.cnn', e1)
    for kind, hfr in (('human', S['HFR'][0]), ('object', S['HFR'][1]), ('geometry', S['HFR'][2])):
        rows(kind + '_bd_embedding_mlp.0', hfr)

    # ---- messages, both levels (every form: sender-only / receiver-specific / relational; concat / bilinear scores)
    E_of = {'h': H, 'o': O, 's': 1}
    feat = {'h': HUMv[:, :2 * h].reshape(nF, H, 2 * h), 'o': OBJv[:, :2 * h].reshape(nF, O, 2 * h),
            's': GEOv[:, :2 * h].reshape(nF, 1, 2 * h)}
    mask_f = objects_mask.view(bs, 1, O).expand(bs, T, O).reshape(nF, O)
    general_f = S.get('frame_general') or {}
    for rel in ops._FRAME_RELS:
        if not getattr(plan, 'rel_' + rel):
            continue
        rk, sk = ops._REL_ENDS[rel]
        agg = general_f.get(rel, {}).get('agg')
        items += _relation_layers(plan, P, ops, rel, False, feat[rk], feat[sk], mask_f if sk == 'o' else None, agg)
    if plan.msg_segment and T > 1:
        sb = S['seg_bufs']
        slots = sb.get('general_slots') or {}
        hs = {'h': sb['hs_h'], 'o': sb['hs_o']}
        # instances = (direction, clip, step) with a previous state: direction 0 step t reads hs[:, t-1, :, :h], direction 1
        # reads hs[:, t+1, :, h:] (the chain start reads zeros: its pre-activation is the bias alone)
        prev = {k: torch.cat([v[:, :T - 1, :, :h].reshape(-1, E_of[k], h), v[:, 1:, :, h:].reshape(-1, E_of[k], h)], 0)
                for k, v in hs.items()}
        mask_s = objects_mask.view(bs, 1, O).expand(bs, T - 1, O).reshape(-1, O).repeat(2, 1)
        for rel in ops._SEG_RELS:
            if not getattr(plan, 'rel_' + rel):
                continue
            rk, sk = ops._REL_ENDS[rel]
            agg = None
            if plan.relational and (0, rel, 'agg') in slots:   # [bs][T][R][h] per direction, every step
                agg = torch.cat([slots[(d, rel, 'agg')].reshape(-1, h) for d in range(2)], 0)
            items += _relation_layers(plan, P, ops, rel, True, prev[rk], prev[sk], mask_s if sk == 'o' else None, agg)

    # ---- hidden layers of the gate networks (discrete_networks_num_layers > 1, vhoi/models.py:523-548)
    for kind, Ev, cols, mlp in (('h', HUMv, plan.gate_cols_h(), 'update_human_segment_mlp'),
                                ('o', OBJv, plan.gate_cols_o(), 'update_object_segment_mlp')):
        g = S['gates'].get(kind) or {}
        acts = g.get('acts') or []
        if not acts or g.get('alias'):
            continue
        rows(mlp + '.0', torch.cat([Ev[:, c:c + h] for c in cols], 1))
        for layer in range(1, len(acts)):
            rows(f'{mlp}.{2 * layer}', acts[layer - 1])

    # ---- position features (embedding style): relu(w s + b) of one scalar per (clip, frame, entity)
    if not plan.periodic:
        for name, scal in (S.get('pos') or {}).items():
            mlp = 'segment_length_mlp.0' if name == 'seglen' else 'time_position_mlp.0'
            for s_k in scal.values():
                rows(mlp, s_k.reshape(-1, 1))
    return [i for i in items if i.rows > 0 and i.units > 0]


def boundary_layers(model, out, width=4.0):
    """out: the list a TGGCN forward returned (its grad_fn carries the saved state). Returns
    {layer name (as in the state_dict, without '.weight'): tensor of unit indices with boundary rows}."""
    found = {}
    with torch.no_grad():
        for it in _layers(model, out):
            idx = _near(*it.pre_mag(), width * it.factor).nonzero().flatten().cpu()
            if len(idx):   # (a layer can appear more than once: both levels of share_level_mlps, several position features)
                found[it.layer] = torch.unique(torch.cat([found[it.layer], idx])) if it.layer in found else idx
    return found


# what the last condition_case() looked at: ReLU units (layer outputs) and activations (units x rows) of the covered layers
# and the (unit, row) activations found inside the rounding band over all rounds
LAST_TOTALS = dict(units=0, activations=0, boundary_activations=0)


def condition_case(model, forward, width=8.0, max_rounds=8):
    """Moves a test case OFF the ReLU rounding boundaries instead of excusing what they do to a gradient comparison:
    runs `forward()` (the HIP path, train mode) and, while some ReLU unit has a row whose pre-activation lies within
    `width` * u * sum|x w| of zero, adds a few band widths to that unit's bias -- the case is re-seeded in exactly the
    coordinates that made it ambiguous, everything else stays. Afterwards oracle and kernels take the same side of
    every ReLU however they order their sums, and the comparison needs no escape clause.
    Returns (number of rounds, {layer: number of units nudged}); raises if the case does not settle."""
    P = dict(model.named_parameters())
    nudged = {}
    for rnd in range(max_rounds):
        out = forward()
        todo = []
        LAST_TOTALS.update(units=0, activations=0)
        if rnd == 0:
            LAST_TOTALS['boundary_activations'] = 0
        with torch.no_grad():
            for it in _layers(model, out):
                LAST_TOTALS['units'] += it.units
                LAST_TOTALS['activations'] += it.units * it.rows
                count, band = _near(*it.pre_mag(), width * it.factor, with_band=True)
                idx = count.nonzero().flatten()
                if len(idx):
                    todo.append((it.layer, idx, band[idx]))
                    LAST_TOTALS['boundary_activations'] += int(count.sum())
        del out
        if not todo:
            return rnd, nudged
        with torch.no_grad():
            for layer, idx, band in todo:
                bias = P.get(layer + '.bias')
                assert bias is not None, ('a boundary unit in a layer without bias: pick another seed', layer)
                bias[idx] += (4.0 * band).to(bias.dtype)
                nudged[layer] = nudged.get(layer, 0) + len(idx)
    raise AssertionError(('the case did not settle off the ReLU boundaries', nudged))
