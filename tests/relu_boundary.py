"""Test infrastructure: which ReLU layers of a forward pass have units sitting ON the boundary, i.e. a pre-activation
whose magnitude is below the fp32 rounding error of its own dot product, so that the summation order alone decides on
which side of zero it lands (CPU oracle and HIP kernels sum in different orders).

A gradient comparison may tolerate a deviation only where this module finds such a unit: the owning layer's gradient
moves in that unit's row, the gradients UPSTREAM of the layer are perturbed, everything downstream and every output
stays put (tools/parity_fuzz.py::_is_owner_or_upstream). The pre-activations are recomputed in fp64 (torch on the
device is the checker here) from the activations the HIP path itself saved for its backward pass."""
import torch

U = 2.0 ** -24   # fp32 unit round-off


def _boundary_units(x, w, b, width=4.0):
    """x (rows, K), w (N, K), b (N,) or None -> per output unit, the number of rows whose pre-activation is within
    `width` * u * sum_k |x_k w_k| of zero (a few units of the typical error of a K-term fp32 dot product, far below
    the worst-case K u bound)."""
    x, w = x.double(), w.double()
    pre = x @ w.t()
    mag = x.abs() @ w.abs().t()
    if b is not None:
        pre = pre + b.double()
        mag = mag + b.double().abs()
    near = (pre.abs() <= width * U * mag) & (mag > 0)
    return near.sum(0)   # (N,)


def boundary_layers(model, out):
    """out: the list a TGGCN forward returned (its grad_fn carries the saved state). Returns
    {layer name (as in the state_dict, without '.weight'): tensor of unit indices with boundary rows}."""
    from twog_gcn_amd import ops
    node = next(o.grad_fn for o in out if o.grad_fn is not None)
    S, plan, P = ops.saved_state(node), node.plan, dict(model.named_parameters())
    x_human, x_objects, _ = node.inputs
    bs, T, H, O, N, h = plan.bs, plan.T, plan.H, plan.O, plan.N, plan.h
    nF = bs * T
    HUMv, OBJv, GEOv = (S[k].view(-1, S[k].shape[-1]) for k in ('HUM', 'OBJ', 'GEO'))
    found = {}

    def check(layer, x):
        w = P.get(layer + '.weight')
        if w is None or x is None or x.shape[-1] != w[0].numel():
            return   # a configuration without this layer / with another operand layout (general relations): not covered
        units = _boundary_units(x.reshape(-1, x.shape[-1]), w.view(w.shape[0], -1), P.get(layer + '.bias'))
        idx = units.nonzero().flatten()
        if len(idx):
            found[layer] = idx.cpu()

    with torch.no_grad():
        check('human_embedding_mlp.0', x_human.view(nF * H, -1)[:, :2048])
        if O:
            check('object_embedding_mlp.0', x_objects.view(nF * O, -1))
        check('geometry_embedding_mlp.0', S['Gout'].view(nF, 128 * N))
        check('geometry_embedding_mlp.2', S.get('t1'))
        check('geometry_embedding_gcn.joint_embed.cnn.3.cnn', S.get('e1'))
        for kind, Ev, hfr in (('human', HUMv, S['HFR'][0]), ('object', OBJv, S['HFR'][1]), ('geometry', GEOv, S['HFR'][2])):
            check(kind + '_bd_embedding_mlp.0', hfr)
        for Ev, rels in ((HUMv, plan.snd_h), (OBJv, plan.snd_o), (GEOv, plan.snd_s)):
            for rel in rels:
                check(ops._FRAME_MLP[rel] + '.0', Ev[:, :2 * h])
        if plan.msg_segment and T > 1 and 'seg_rels' in S:
            sb = S['seg_bufs']
            for rels, hs in ((S['seg_rels'][0], sb['hs_h']), (S['seg_rels'][1], sb['hs_o'])):
                prev = torch.cat([hs[:, :T - 1, :, :h].reshape(-1, h), hs[:, 1:, :, h:].reshape(-1, h)], 0)
                for rel in rels:
                    check(ops._SEG_MLP[rel] + '.0', prev)
    return found
