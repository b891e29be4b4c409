"""Shared test helpers: load golden fixtures, rebuild their deterministic weights."""
import json
import os

import numpy as np
import torch

from oracle import detgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

G4_CASES = ['c2_stage1', 'c2_stage2', 'c1_stage1', 'c1_stage2', 'c5_stage1', 'c2_geo2human', 'c2_dot_st']
# round 2: the rest of the constructor's configuration surface (reference outputs, tools/make_golden.py)
G4_CASES_R2 = ['c1_sah', 'c1_coh', 'c2_gate3', 'c2_nobias', 'c2_time_s', 'c2_time_u_periodic', 'c2_seglen',
               'c2_seglen_periodic', 'c2_concat', 'c2_general', 'c2_specific', 'c2_relational', 'c2_distance',
               'c2_ctor_defaults', 'c1_relational_geo2h', 'c2_concat_f', 'c2_general_f', 'c2_specific_f',
               'c2_specific_concat_mp_f', 'c2_specific_general_f', 'c2_relational_f', 'c2_distance_f',
               'c1_relational_geo2h_f']
# every one of them runs on the product path (HIP kernels / their test double)
G4_CASES_R2_BUILT = list(G4_CASES_R2)


def load_g4(name):
    z = np.load(os.path.join(GOLDEN, f'g4_{name}.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta_json']))
    return z, meta


def det_state_dict(shapes: dict, seed: int, gain: float = 1.0, requires_grad: bool = False):
    vals = detgen.fill_state_dict({k: tuple(v) for k, v in shapes.items()}, seed=seed, gain=gain)
    sd = {}
    for k, v in vals.items():
        t = torch.from_numpy(np.asarray(v)).clone()
        if requires_grad and t.is_floating_point() and 'running_' not in k:
            t.requires_grad_(True)
        sd[k] = t
    return sd


def g4_inputs(z):
    kw = dict(x_human=torch.from_numpy(z['x_human']), x_objects=torch.from_numpy(z['x_objects']),
              objects_mask=torch.from_numpy(z['objects_mask']))
    bs, T = z['x_human'].shape[:2]
    kw['steps_per_example'] = torch.full((bs,), float(T))
    if 'human_segmentation' in z.files:
        kw['human_segmentation'] = torch.from_numpy(z['human_segmentation'])
    if 'objects_segmentation' in z.files:
        kw['objects_segmentation'] = torch.from_numpy(z['objects_segmentation'])
    for k in ('human_human_distances', 'human_object_distances', 'object_object_distances'):
        if k in z.files:
            kw[k] = torch.from_numpy(z[k])
    return kw


def sample_grad(g: torch.Tensor, limit=4096) -> np.ndarray:
    flat = g.detach().flatten().cpu().numpy()
    if flat.size <= limit:
        return flat.copy()
    stride = flat.size // limit
    return flat[::stride][:limit].copy()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ---- G12 (training trajectory): inputs, targets and initial weights are closed-form (oracle/detgen.py); the fixture holds
# only what the reference produced. tools/make_golden.py generates the fixture from these same two functions.
def synth_inputs(name, H, O, N, bs, T, seed):
    """x_human (bs,T,H,2048+4N), x_objects (bs,T,O,2048), objects_mask (bs,O) -- the generator of the G4 / G12 inputs."""
    vis = np.maximum(detgen.normal(name + '.xh', (bs, T, H, 2048), seed=seed), 0.0)
    pos = detgen.uniform(name + '.pos', (bs, T, N, 2), 0.0, 1.0, seed=seed)
    vel = detgen.normal(name + '.vel', (bs, T, N, 2), std=0.5, seed=seed)
    geo = np.concatenate([pos, vel], axis=-1).reshape(bs, T, 1, 4 * N)
    geo = np.repeat(geo, H, axis=2)
    x_human = np.concatenate([vis, geo], axis=-1).astype(np.float32)
    x_objects = np.maximum(detgen.normal(name + '.xo', (bs, T, O, 2048), seed=seed), 0.0).astype(np.float32)
    mask = np.ones((bs, O), dtype=np.float32)
    mask[0, O - 1] = 0.0
    if bs > 1 and O > 2:
        mask[1, O - 2:] = 0.0
    x_objects = x_objects * mask[:, None, :, None]
    return x_human, x_objects, mask


def g12_targets(meta, step):
    """Targets of training step `step`: class labels with a ragged tail (ignore_index -1), gate targets."""
    bs, T, H, seed, n_cls = meta['bs'], meta['T'], meta['H'], meta['seed'], meta['classes'][0]
    cls = [(detgen.uniform01(f'g12.s{step}.cls{i}', (bs, T, H), seed=seed) * n_cls).astype(np.int64) for i in range(2)]
    seg = (detgen.uniform01(f'g12.s{step}.seg', (bs, T, H), seed=seed) > 0.55).astype(np.float32)
    for a in cls:
        a[1, T - 2:] = -1
    seg[1, T - 2:] = -1.0
    return cls, seg


def load_g12():
    z = np.load(os.path.join(GOLDEN, 'g12_training_trajectory.npz'), allow_pickle=False)
    return z, json.loads(str(z['meta_json']))


def g12_step_batch(meta, step):
    """(model kwargs, criterion targets) of training step `step` as torch tensors."""
    xh, xo, mask = synth_inputs(f'g12.s{step}', meta['H'], meta['O'], meta['N'], meta['bs'], meta['T'], meta['seed'])
    cls, seg = g12_targets(meta, step)
    kw = dict(x_human=torch.from_numpy(xh), x_objects=torch.from_numpy(xo), objects_mask=torch.from_numpy(mask),
              steps_per_example=torch.full((meta['bs'],), float(meta['T'])))
    target = [torch.from_numpy(seg), torch.from_numpy(seg)] + [torch.from_numpy(cls[i % 2]) for i in range(4)]
    return kw, target
