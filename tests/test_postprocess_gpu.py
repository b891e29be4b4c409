"""GPU: twog_predict_labels / twog_f1_at_k through the C ABI against the reference's golden vector G8 (bit-exact labels,
F1 within fp32 rounding), against the oracle on seeded random sequences, and at bench size by a size-independent property
(predictions equal to the targets give F1 = 1 for every overlap)."""
import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import postprocess as pp
from oracle import postprocess_ref as R
from tests.test_postprocess_cpu import check_mirror_against_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_golden_g8():
    check_mirror_against_golden(DEV)


@pytest.mark.parametrize('n_seq,n_steps,ncls,seed', [(37, 120, 13, 0), (5, 1, 3, 1), (64, 333, 10, 2), (3, 17, 1, 3)])
def test_random_sequences_vs_oracle(n_seq, n_steps, ncls, seed):
    rng = np.random.RandomState(seed)
    runs = lambda: np.repeat(rng.randint(0, ncls + 1, size=n_steps), rng.randint(1, 6, size=n_steps))[:n_steps]
    yt = np.stack([runs() for _ in range(n_seq)]).astype(np.int64)
    yp = np.stack([runs() for _ in range(n_seq)]).astype(np.int64)
    yp[: n_seq // 2] = yt[: n_seq // 2]
    yp[: n_seq // 2, ::7] = (yp[: n_seq // 2, ::7] + 1) % (ncls + 1)
    yt[rng.rand(n_seq, n_steps) < 0.1] = -1
    if n_seq > 4:
        yt[1] = -1
    for ov in (0.1, 0.25, 0.5, 1.0):
        want = R.f1_at_k(yt, yp, ncls, ov, ignore_value=-1.0)
        got = pp.f1_at_k(torch.from_numpy(yt).to(DEV), torch.from_numpy(yp).to(DEV), ncls, overlap=ov, ignore_value=-1.0)
        assert abs(got - want) < 1e-6, (ov, got, want)
    want = R.f1_at_k(np.abs(yt), yp, ncls, 0.25, ignore_value=None)
    got = pp.f1_at_k(torch.from_numpy(np.abs(yt)).to(DEV), torch.from_numpy(yp).to(DEV), ncls, overlap=0.25)
    assert abs(got - want) < 1e-6


def test_bench_size_properties():
    bs, C, T, E, ds = 64, 13, 120, 2, 3
    g = torch.Generator().manual_seed(0)
    logp = torch.log_softmax(torch.randn(bs, C, T, E, generator=g), 1).to(DEV)
    tgt = torch.zeros(bs, T * ds + 2, E, dtype=torch.int64, device=DEV)
    lab = pp.predict_labels(logp, tgt, ds)
    assert lab.shape == (bs, T * ds + 2, E)
    want = torch.repeat_interleave(logp.argmax(1), ds, dim=1)
    assert torch.equal(lab[:, :T * ds], want) and torch.equal(lab[:, -1], want[:, -1]) and torch.equal(lab[:, -2], want[:, -1])
    seqs = lab.transpose(1, 2).reshape(-1, lab.shape[1])
    for ov in (0.1, 0.5, 1.0):
        assert pp.f1_at_k(seqs, seqs, C, overlap=ov, ignore_value=-1.0) == 1.0
