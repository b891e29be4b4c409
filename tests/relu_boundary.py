"""Test infrastructure: which ReLU layers of a forward pass have units sitting ON the boundary, i.e. a pre-activation
whose magnitude is below the fp32 rounding error of its own dot product, so that the summation order alone decides on
which side of zero it lands (CPU oracle and HIP kernels sum in different orders).

A gradient comparison may tolerate a deviation only where this module finds such a unit: the owning layer's gradient
moves in that unit's row, the gradients UPSTREAM of the layer are perturbed, everything downstream and every output
stays put (tools/parity_fuzz.py::_is_owner_or_upstream). The pre-activations are recomputed in fp64 (torch on the
device is the checker here) from the activations the HIP path itself saved for its backward pass."""
import torch

U = 2.0 ** -24   # fp32 unit round-off


def _boundary_units(x, w, b, width=4.0, with_band=False):
    """x (rows, K), w (N, K), b (N,) or None -> per output unit, the number of rows whose pre-activation is within
    `width` * u * sum_k |x_k w_k| of zero (a few units of the typical error of a K-term fp32 dot product, far below
    the worst-case K u bound). with_band: also the widest such band per unit (what a bias nudge has to clear)."""
    x, w = x.double(), w.double()
    pre = x @ w.t()
    mag = x.abs() @ w.abs().t()
    if b is not None:
        pre = pre + b.double()
        mag = mag + b.double().abs()
    band = width * U * mag
    near = (pre.abs() <= band) & (mag > 0)
    if with_band:
        return near.sum(0), torch.where(near, band, torch.zeros_like(band)).max(0).values
    return near.sum(0)   # (N,)


# the first 1x1 convolution of the geometric-level GCN reads the BatchNorm output: the two implementations' batch
# statistics differ by rounding (fp64 ordered sums vs torch's), which moves its pre-activations by more than the
# rounding of its own 4-term dot product -- its band is this many times wider
GCN_CONV1_WIDTH_FACTOR = 16.0


def _layers(model, out):
    """Yields (layer name, input rows (rows, K), weight (N, K), bias or None, width factor) for every ReLU layer of the
    forward pass whose input the HIP path saved."""
    from twog_gcn_amd import ops
    node = next(o.grad_fn for o in out if o.grad_fn is not None)
    S, plan, P = ops.saved_state(node), node.plan, dict(model.named_parameters())
    x_human, x_objects, _ = node.inputs
    bs, T, H, O, N, h = plan.bs, plan.T, plan.H, plan.O, plan.N, plan.h
    nF = bs * T
    HUMv, OBJv, GEOv = (S[k].view(-1, S[k].shape[-1]) for k in ('HUM', 'OBJ', 'GEO'))

    def item(layer, x, factor=1.0):
        w = P.get(layer + '.weight')
        if w is None or x is None or x.shape[-1] != w[0].numel():
            return None  # a configuration without this layer / with another operand layout (general relations): not covered
        return layer, x.reshape(-1, x.shape[-1]), w.view(w.shape[0], -1), P.get(layer + '.bias'), factor

    items = [item('human_embedding_mlp.0', x_human.view(nF * H, -1)[:, :2048])]
    if O:
        items.append(item('object_embedding_mlp.0', x_objects.view(nF * O, -1)))
    items.append(item('geometry_embedding_mlp.0', S['Gout'].view(nF, 128 * N)))
    items.append(item('geometry_embedding_mlp.2', S.get('t1')))
    if S.get('ab') is not None:
        # x^[f, n, c] = a[c*N + n] * x[f, n, c] + b[c*N + n] (BatchNorm folded, models_gcn.py:45-50), geometry of human 0
        ab = S['ab'].view(2, 4, N).double()
        xg = x_human.view(nF, H, -1)[:, 0, 2048:].reshape(nF, N, 4).double()
        xhat = xg * ab[0].t().unsqueeze(0) + ab[1].t().unsqueeze(0)
        items.append(item('geometry_embedding_gcn.joint_embed.cnn.1.cnn', xhat.reshape(nF * N, 4), GCN_CONV1_WIDTH_FACTOR))
        # the second convolution's input e1 = relu(W1 x^ + b1) is not stored by the fused forward kernel: recomputed here
        c1 = 'geometry_embedding_gcn.joint_embed.cnn.1.cnn'
        w1, b1 = P[c1 + '.weight'].view(64, 4).double(), P[c1 + '.bias'].double()
        e1 = torch.relu(xhat.reshape(nF * N, 4) @ w1.t() + b1)
        items.append(item('geometry_embedding_gcn.joint_embed.cnn.3.cnn', e1))
    for kind, hfr in (('human', S['HFR'][0]), ('object', S['HFR'][1]), ('geometry', S['HFR'][2])):
        items.append(item(kind + '_bd_embedding_mlp.0', hfr))
    for Ev, rels in ((HUMv, plan.snd_h), (OBJv, plan.snd_o), (GEOv, plan.snd_s)):
        for rel in rels:
            items.append(item(ops._FRAME_MLP[rel] + '.0', Ev[:, :2 * h]))
    if plan.msg_segment and T > 1 and 'seg_rels' in S:
        sb = S['seg_bufs']
        for rels, hs in ((S['seg_rels'][0], sb['hs_h']), (S['seg_rels'][1], sb['hs_o'])):
            prev = torch.cat([hs[:, :T - 1, :, :h].reshape(-1, h), hs[:, 1:, :, h:].reshape(-1, h)], 0)
            for rel in rels:
                items.append(item(ops._SEG_MLP[rel] + '.0', prev))
    return [i for i in items if i is not None and i[1].shape[0] > 0]


def boundary_layers(model, out, width=4.0):
    """out: the list a TGGCN forward returned (its grad_fn carries the saved state). Returns
    {layer name (as in the state_dict, without '.weight'): tensor of unit indices with boundary rows}."""
    found = {}
    with torch.no_grad():
        for layer, x, w, b, factor in _layers(model, out):
            idx = _boundary_units(x, w, b, width * factor).nonzero().flatten()
            if len(idx):
                found[layer] = idx.cpu()
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
            for layer, x, w, b, factor in _layers(model, out):
                LAST_TOTALS['units'] += int(w.shape[0])
                LAST_TOTALS['activations'] += int(w.shape[0]) * int(x.shape[0])
                count, band = _boundary_units(x, w, b, width * factor, with_band=True)
                idx = count.nonzero().flatten()
                if len(idx):
                    todo.append((layer, idx, band[idx]))
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
