"""CPU: the oracle (oracle/cpu_ref.py) against golden vectors captured from the real reference
(tools/make_golden.py). This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, detgen
from tests.helpers import GOLDEN, G4_CASES, G4_CASES_R2, load_g4, det_state_dict, g4_inputs, sample_grad, rel_err

TOL = 2e-5  # oracle vs reference: same fp32 CPU ops, differences are summation order only


@pytest.mark.parametrize('N', [19, 26, 30, 34])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_g1_geo_gcn(N, mode):
    z = np.load(f'{GOLDEN}/g1_geo_gcn.npz')
    key = f'N{N}_{mode}'
    shapes = {k[len(key) + 6:]: z[k].shape for k in z.files if k.startswith(key + '_grad_')}
    shapes.update({'joint_embed.cnn.0.bn.running_mean': (4 * N,), 'joint_embed.cnn.0.bn.running_var': (4 * N,),
                   'joint_embed.cnn.0.bn.num_batches_tracked': ()})
    sd = det_state_dict(shapes, seed=100 + N, requires_grad=True)
    sd = {'g.' + k: v for k, v in sd.items()}
    bs, T = 2, 5
    x = torch.from_numpy(detgen.normal(f'g1.x.{N}', (bs, 4, N, T), std=1.0, seed=1))
    r = torch.from_numpy(detgen.normal(f'g1.r.{N}', (bs, 128, N, T), std=1.0, seed=2))
    bn_state = {}
    y = cpu_ref.geo_gcn(sd, x, training=(mode == 'train'), prefix='g', bn_state=bn_state)
    assert rel_err(y.detach().numpy(), z[key + '_y']) < TOL
    (y * r).sum().backward()
    for k in z.files:
        if k.startswith(key + '_grad_'):
            name = 'g.' + k[len(key) + 6:]
            g = sd[name].grad.numpy()  # (s2 bias grad is analytically 0: row-constant logit term)
            assert np.abs(g - z[k]).max() < 5e-5 * np.abs(z[k]).max() + 2e-6, name
    if mode == 'train':
        assert rel_err(bn_state['running_mean'].numpy(), z[key + '_running_mean']) < TOL
        assert rel_err(bn_state['running_var'].numpy(), z[key + '_running_var']) < TOL
        assert int(bn_state['num_batches_tracked']) == int(z[key + '_nbt'])


def test_g3_messages_and_attention():
    z = np.load(f'{GOLDEN}/g3_messages.npz')
    q, keys, mask = (torch.from_numpy(z[k]) for k in ('q', 'keys', 'mask'))
    d, hm = q.shape[1], 6
    sd = {}
    sd.update({'m1.' + k: v for k, v in det_state_dict({'0.weight': (hm, d), '0.bias': (hm,)}, 31).items()})
    sd.update({'m2.' + k: v for k, v in det_state_dict({'0.weight': (hm, 2 * d), '0.bias': (hm,)}, 32).items()})
    sd.update({'a1.' + k: v for k, v in det_state_dict({'0.weight': (1, 2 * d), '0.bias': (1,)}, 33).items()})
    sd.update({'a4.' + k: v for k, v in det_state_dict({'weight': (1, d, d), 'bias': (1,)}, 34).items()})
    assert rel_err(cpu_ref.non_relational_message(sd, q, keys, mask, 'v1', 'm1').numpy(), z['msg_v1']) < TOL
    assert rel_err(cpu_ref.non_relational_message(sd, q, keys, mask, 'v2', 'm2').numpy(), z['msg_v2']) < TOL
    for style, fn in (('v1', 'a1'), ('v2', None), ('v3', None), ('v4', 'a4')):
        w = cpu_ref.attention_weights(sd, q, keys, mask, style, fn).numpy()
        assert not np.isnan(w).any()
        assert np.abs(w - z['att_' + style]).max() < 1e-6, style
    assert np.all(z['att_v3'][2] == 0.0)  # fully masked row: NaN -> 0


def test_g5_reorder_and_filter():
    z = np.load(f'{GOLDEN}/g5_reorder_filter.npz')
    out = cpu_ref.reorder_hidden_states(torch.from_numpy(z['hx']), torch.from_numpy(z['ux']))
    assert np.array_equal(out.numpy(), z['reordered'])
    soft = torch.from_numpy(z['soft'])
    for thr in (0.1, 0.5):
        f = torch.stack(cpu_ref.filter_soft_decisions([s for s in soft], thr), 0).numpy()
        assert np.array_equal(f, z[f'filtered_{thr}'])


@pytest.mark.parametrize('name', G4_CASES + G4_CASES_R2)
def test_g4_full_forward_backward(name):
    z, meta = load_g4(name)
    sd = det_state_dict(meta['state_dict_shapes'], seed=meta['seed'], gain=meta['gain'], requires_grad=True)
    kw = g4_inputs(z)
    noise = torch.from_numpy(z['gumbel_noise'])
    aux = {}
    out = cpu_ref.tggcn_forward(sd, meta['cfg'], training=True, gumbel_noise=noise if len(noise) else None,
                                aux=aux, **kw)
    n_out = len([k for k in z.files if k.startswith('out')])
    assert len(out) == n_out
    for i, o in enumerate(out):
        ref = z[f'out{i}']
        assert tuple(o.shape) == ref.shape
        if ref.ndim == 3 and i < n_out - 4 and np.all((ref == 0) | (ref == 1)):  # hard gates: exact
            assert np.array_equal(o.detach().numpy(), ref), f'out{i}'
        else:
            assert np.abs(o.detach().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), f'out{i}'
    assert rel_err(aux['bn_state']['running_mean'].numpy(), z['bn_running_mean']) < TOL
    assert rel_err(aux['bn_state']['running_var'].numpy(), z['bn_running_var']) < TOL
    if not bool(z['backward_ok']):
        return
    loss = 0
    for i, o in enumerate(out):
        if o.requires_grad:
            r = torch.from_numpy(detgen.normal(f'{name}.r{i}', tuple(o.shape), seed=meta['seed']))
            loss = loss + (o * r).sum()
    assert abs(float(loss) - float(z['loss'])) < 1e-3 * max(1.0, abs(float(z['loss'])))
    loss.backward()
    none_ref = set(z['none_grads'].tolist())
    for pname, p in sd.items():
        if not p.requires_grad:
            continue
        if pname in none_ref:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname  # dead parameter (Appendix A6)
            continue
        g_ref = z['grad_' + pname]
        g = sample_grad(p.grad)
        scale = max(np.abs(g_ref).max(), 1e-6)
        assert np.abs(g - g_ref).max() < 2e-4 * scale + 1e-6, (pname, np.abs(g - g_ref).max(), scale)


def test_g7_losses():
    z = np.load(f'{GOLDEN}/g7_losses.npz')
    for ds, n_out in (('mphoi', 6), ('cad120', 12)):
        outs = [torch.from_numpy(z[f'{ds}_o{i}']) for i in range(n_out)]
        tgts = [torch.from_numpy(z[f'{ds}_t{i}']) for i in range(n_out)]
        if ds == 'cad120':
            w = [0.5, 0.25, 0.7, 0.7] + [0.3] * 4 + [1.0, 1.0, 1.0, 1.0]
        else:
            w = [0.5, 0.7] + [0.3] * 2 + [1.0, 1.0]
        got = [float(v) for v in cpu_ref.loss_list(outs, tgts, w, cad120=(ds == 'cad120'))]
        assert np.allclose(got, z[f'{ds}_losses'], rtol=1e-5, atol=1e-6), (got, z[f'{ds}_losses'])
