"""Every way of writing a parameter must reach the next forward (round 4 kept the packed segment-level
sender-MLP weights per optimizer step behind a (data_ptr, version) stamp, and `p.data.mul_()` -- the EMA / clipping /
re-initialisation idiom, and the reference's own init style, pyrutils/torch/models_gcn.py:28 `self.weight.data.uniform_` --
bumps no version counter: the product's outputs did not move while the oracle's moved by 1.1).

Sequence on one model: forward -> p.data.mul_() -> p.data.copy_() -> load_state_dict -> torch.optim.Adam.step() (what
train.py:38-39 does) -> FlatParameters re-homing + FusedAdam.step() -> a raw write into the flat buffer; after EACH write the
forward must equal the oracle evaluated on the model's CURRENT state_dict (1e-4 relative, the bar of north_star) AND must
have moved by what the oracle moved. Runs on the kernel test double (CPU) and on the HIP kernels (GPU)."""
import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import kernels as twog_kernels
from twog_gcn_amd import ops
from twog_gcn_amd.models import TGGCN
from oracle import cpu_ref
from tests.helpers import load_g4, det_state_dict, g4_inputs

TOL = 1e-4


def _run(device):
    z, meta = load_g4('c2_stage1')          # stage-1 MPHOI layout, message_segment on: the packed sender MLPs are in use
    N = meta['N']
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=tuple(meta['classes']), **meta['cfg'])
    m.load_state_dict(det_state_dict(meta['state_dict_shapes'], seed=meta['seed'], gain=meta['gain']))
    m = m.to(device).eval()                 # eval: BatchNorm statistics fixed, every forward is a pure function of the weights
    kw = g4_inputs(z)
    noise = torch.from_numpy(z['gumbel_noise'])
    seg_w = [n for n, _ in m.named_parameters() if n.endswith('_segment_message_mlp.0.weight')
             and not n.startswith('geometry')]
    assert len(seg_w) == 4, seg_w
    P = dict(m.named_parameters())

    def product():
        m._gumbel_noise_override = noise if len(noise) else None
        out = m(**{k: v.to(device) for k, v in kw.items()})
        return [o.detach().cpu() for o in out]

    def oracle():
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        out = cpu_ref.tggcn_forward(sd, dict(m.cfg), kw['x_human'], kw['x_objects'], kw['objects_mask'],
                                    human_segmentation=kw.get('human_segmentation'),
                                    objects_segmentation=kw.get('objects_segmentation'), training=False,
                                    gumbel_noise=noise if len(noise) else None, steps_per_example=kw['steps_per_example'])
        return [o.detach() for o in out]

    def check(what, prev):
        got, want = product(), oracle()
        for i, (g, w) in enumerate(zip(got, want)):
            err = float((g - w).abs().max())
            assert err < TOL * max(1.0, float(w.abs().max())), (what, i, err)
        if prev is not None:   # the write really changed the function, and the product followed
            moved_o = max(float((a - b).abs().max()) for a, b in zip(want[2:], prev[1][2:]))
            moved_p = max(float((a - b).abs().max()) for a, b in zip(got[2:], prev[0][2:]))
            assert moved_o > 1e-3, (what, 'the write did not change the oracle', moved_o)
            assert abs(moved_p - moved_o) < 0.05 * moved_o + 1e-4, (what, moved_p, moved_o)
        return got, want

    with torch.no_grad():
        st = check('initial', None)
        for n in seg_w:                                     # in place through .data: no version bump
            P[n].data.mul_(-3.0)
        st = check('p.data.mul_', st)
        for i, n in enumerate(seg_w):
            P[n].data.copy_(torch.roll(P[n].data, 1 + i, 0) * 0.5)
        st = check('p.data.copy_', st)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        for n in seg_w:
            sd[n] = sd[n] * -0.7 + 0.01
        m.load_state_dict(sd)
        st = check('load_state_dict', st)

    def one_training_step(opt_step, zero):
        m.train()
        zero()
        m._gumbel_noise_override = noise if len(noise) else None
        out = m(**{k: v.to(device) for k, v in kw.items()})
        sum((o * o).sum() for o in out if o.requires_grad).backward()
        opt_step()
        m.eval()

    # restore running statistics after each train-mode forward: the check below is about the weights
    bn = m.geometry_embedding_gcn.joint_embed.cnn[0].bn
    bn_state = {k: v.clone() for k, v in bn.state_dict().items()}
    opt = torch.optim.Adam(m.parameters(), lr=2e-2)
    one_training_step(opt.step, opt.zero_grad)
    bn.load_state_dict(bn_state)
    with torch.no_grad():
        st = check('torch.optim.Adam.step', st)

    from twog_gcn_amd.distributed import DataParallel, FusedAdam
    dp = DataParallel(m)                                     # parameters re-homed into one flat buffer
    with torch.no_grad():
        st2 = check('FlatParameters re-homing', None)        # same function as before the move
        for a, b in zip(st2[0], st[0]):
            assert float((a - b).abs().max()) < 1e-5
    fopt = FusedAdam(dp.flat, lr=2e-2)
    one_training_step(lambda: fopt.step(dp.grad_scale), dp.zero_grad)
    bn.load_state_dict(bn_state)
    with torch.no_grad():
        st = check('FusedAdam.step', st)
        dp.flat.flat.mul_(0.9)                               # a raw write into the flat buffer (nobody is told)
        st = check('flat buffer write', st)
    dp.close()


def test_parameter_writes_reach_the_next_forward_on_the_kernel_test_double():
    from tests.fake_kernels import FakeKernels
    twog_kernels._set_backend_for_tests(FakeKernels())
    try:
        _run('cpu')
    finally:
        twog_kernels._set_backend_for_tests(None)


@pytest.mark.gpu
def test_parameter_writes_reach_the_next_forward_on_the_hip_kernels():
    twog_kernels._set_backend_for_tests(None)
    assert twog_kernels.get_kernels().name == 'hip'
    _run('cuda:0')
