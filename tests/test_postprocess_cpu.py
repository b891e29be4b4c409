"""CPU: the oracle of the inference post-processing against the golden vector G8 recorded from the reference
(predict.py match_shape / argmax, pyrutils.metrics.f1_at_k), and the host mirror (2g-gcn_amd/postprocess.py) through the
kernel-interface test double."""
import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import kernels as twog_kernels
from twog_gcn_amd import postprocess as pp
from tests.fake_kernels import FakeKernels
from tests.helpers import GOLDEN
from oracle import postprocess_ref as R

OVERLAPS = (0.1, 0.25, 0.5)


@pytest.fixture()
def fake_backend():
    twog_kernels._set_backend_for_tests(FakeKernels())
    yield
    twog_kernels._set_backend_for_tests(None)


def g8():
    return np.load(f'{GOLDEN}/g8_postprocess.npz')


def test_oracle_matches_reference_golden():
    z = g8()
    for ci in z['pl_cases']:
        ds, steps = (int(v) for v in z[f'pl{ci}_cfg'])
        got = R.predict_labels(z[f'pl{ci}_logp'], ds, steps)
        assert np.array_equal(got, z[f'pl{ci}_labels'])
    for fi in range(3):
        ncls = int(z[f'f1_{fi}_ncls'])
        got = [R.f1_at_k(z[f'f1_{fi}_true'], z[f'f1_{fi}_pred'], ncls, ov, ignore_value=-1.0) for ov in OVERLAPS]
        assert np.allclose(got, z[f'f1_{fi}_values'], rtol=0, atol=1e-12), (got, z[f'f1_{fi}_values'])


def check_mirror_against_golden(device):
    z = g8()
    for ci in z['pl_cases']:
        ds, steps = (int(v) for v in z[f'pl{ci}_cfg'])
        logp = torch.from_numpy(z[f'pl{ci}_logp']).to(device)
        tgt = torch.zeros(logp.shape[0], steps, logp.shape[3], dtype=torch.int64, device=device)
        got = pp.predict_labels(logp, tgt, ds)
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), z[f'pl{ci}_labels'])   # bit exact
        # the resized log-probabilities of the reference's own two-step route give the same labels
        ref = pp.match_shape(torch.repeat_interleave(logp, ds, dim=-2), tgt) if ds > 1 else logp
        assert np.array_equal(ref.argmax(1).cpu().numpy(), z[f'pl{ci}_labels'])
    for fi in range(3):
        ncls = int(z[f'f1_{fi}_ncls'])
        yt = torch.from_numpy(z[f'f1_{fi}_true']).to(device)
        yp = torch.from_numpy(z[f'f1_{fi}_pred']).to(device)
        for ov, want in zip(OVERLAPS, z[f'f1_{fi}_values']):
            got = pp.f1_at_k(yt, yp, ncls, overlap=ov, ignore_value=-1.0)
            assert abs(got - want) < 1e-6, (fi, ov, got, want)


def test_mirror_matches_reference_golden(fake_backend):
    check_mirror_against_golden('cpu')


def test_evaluate_f1_at_k_layout(fake_backend):
    z = g8()
    yt, yp = torch.from_numpy(z['f1_1_true']), torch.from_numpy(z['f1_1_pred'])
    n = yt.shape[0] // 2
    targets = {'sub-activity_recognition': yt.view(n, 2, -1).transpose(1, 2)}     # (N, T, E) like process_output
    outputs = {'sub-activity_recognition': yp.view(n, 2, -1).transpose(1, 2)}
    res = pp.evaluate_f1_at_k(targets, outputs, int(z['f1_1_ncls']), None, overlap=0.25)
    assert abs(res['sub-activity_recognition'] - z['f1_1_values'][1]) < 1e-6
