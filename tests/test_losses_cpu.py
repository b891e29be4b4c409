"""CPU: the loss interface mirror (2g-gcn_amd/losses.py: select_loss / multi_task_loss, reference vhoi/losses.py:8-70
and pyrutils/torch/losses.py:7-51) against the golden vector G7 recorded from the reference, and its hand-written
backward against autograd of the oracle -- through the torch test double of the kernel interface."""
import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import kernels as twog_kernels
from twog_gcn_amd import losses
from tests.fake_kernels import FakeKernels
from tests.helpers import GOLDEN
from oracle import cpu_ref

MISC = dict(anticipation_loss_weight=1.0, budget_loss=dict(add=True, human_weight=0.5, object_weight=0.25),
            first_level_loss_weight=0.3, segmentation_loss=dict(add=True, pretrain=False, weight=0.7))


@pytest.fixture()
def fake_backend():
    twog_kernels._set_backend_for_tests(FakeKernels())
    yield
    twog_kernels._set_backend_for_tests(None)


def g7(ds):
    z = np.load(f'{GOLDEN}/g7_losses.npz')
    n = 12 if ds == 'cad120' else 6
    return ([torch.from_numpy(z[f'{ds}_o{i}']) for i in range(n)], [torch.from_numpy(z[f'{ds}_t{i}']) for i in range(n)],
            z[f'{ds}_losses'], [str(s) for s in z[f'{ds}_names']])


def oracle_weights(ds):
    return [0.5, 0.25, 0.7, 0.7] + [0.3] * 4 + [1.0] * 4 if ds == 'cad120' else [0.5, 0.7] + [0.3] * 2 + [1.0, 1.0]


def check_against_golden_and_oracle(ds, device):
    outs, tgts, want, names = g7(ds)
    crit, got_names = losses.select_loss('2G-GCN', 'multiple', ds, dict(misc=MISC))
    assert got_names == names
    xs = [o.clone().to(device).requires_grad_(True) for o in outs]
    got = crit(xs, [t.to(device) for t in tgts])
    assert len(got) == len(want)
    assert np.allclose([float(v.detach()) for v in got], want, rtol=1e-5, atol=1e-6), ([float(v.detach()) for v in got], want)
    sum(got).backward()
    xo = [o.clone().requires_grad_(True) for o in outs]
    sum(cpu_ref.loss_list(xo, tgts, oracle_weights(ds), cad120=(ds == 'cad120'))).backward()
    for a, b in zip(xs, xo):
        assert torch.allclose(a.grad.cpu(), b.grad, rtol=1e-5, atol=1e-7), float((a.grad.cpu() - b.grad).abs().max())


@pytest.mark.parametrize('ds', ['mphoi', 'cad120'])
def test_select_loss_matches_reference_golden(ds, fake_backend):
    check_against_golden_and_oracle(ds, 'cpu')


def test_stage1_weights_skip_backward_of_zero_weight_terms(fake_backend):
    outs, tgts, _, _ = g7('mphoi')
    crit, _ = losses.select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))  # stage-1 defaults
    xs = [o.clone().requires_grad_(True) for o in outs]
    got = crit(xs, tgts)
    assert [float(v.detach()) for v in got[:4]] == [0.0] * 4
    sum(got).backward()
    assert all(x.grad is None for x in xs[:4]) and all(x.grad is not None for x in xs[4:])


def test_all_ignored_and_unsupported(fake_backend):
    x = torch.log_softmax(torch.randn(2, 5, 3, 2), 1).requires_grad_(True)
    t = torch.full((2, 3, 2), -1, dtype=torch.int64)
    v = losses.nll_loss(x, t, ignore_index=-1)
    assert torch.isnan(v)          # torch's mean over an empty selection
    v.backward()
    assert float(x.grad.abs().max()) == 0.0
    p = torch.rand(2, 3, 2)
    assert float(losses.binary_cross_entropy_loss(p, torch.full((2, 3, 2), -1.0))) == 0.0
    assert float(losses.budget_loss(p, torch.full((2, 3, 2), -1.0))) == 0.0
    with pytest.raises(NotImplementedError):
        losses.multi_task_loss([x], [t], [losses.nll_loss], reduction='sum')
    with pytest.raises(NotImplementedError):
        losses.select_loss('cad120_baseline', 'multiple', 'cad120', dict(misc={}))
