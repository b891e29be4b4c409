"""GPU: the fused multi-task loss kernels (twog_multitask_loss_fwd/bwd through the C ABI) against the reference's golden
vector G7, against autograd of the oracle on seeded random inputs with ignored targets, and at bench size through
size-independent properties (run-to-run bit equality; the gradient of an NLL term sums to -weight)."""
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import losses
from oracle import cpu_ref
from tests.test_losses_cpu import check_against_golden_and_oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('ds', ['mphoi', 'cad120'])
def test_golden_g7_and_oracle_gradients(ds):
    check_against_golden_and_oracle(ds, DEV)


@pytest.mark.parametrize('bs,T,E,C', [(3, 17, 2, 13), (1, 1, 1, 2), (5, 40, 5, 10)])
def test_random_terms_vs_oracle(bs, T, E, C):
    g = torch.Generator().manual_seed(bs * 100 + T)
    outs, tgts = [], []
    for i in range(6):
        if i < 2:
            o = torch.rand(bs, T, E, generator=g) * 0.9 + 0.05
            t = (torch.rand(bs, T, E, generator=g) > 0.5).float()
            t[torch.rand(bs, T, E, generator=g) < 0.2] = -1.0
        else:
            o = torch.log_softmax(torch.randn(bs, C, T, E, generator=g), 1)
            t = torch.randint(0, C, (bs, T, E), generator=g)
            t[torch.rand(bs, T, E, generator=g) < 0.2] = -1
        outs.append(o)
        tgts.append(t)
    w = [0.5, 0.7, 0.3, 0.3, 1.0, 2.0]
    fns = (losses.budget_loss, losses.binary_cross_entropy_loss) + (losses.nll_loss,) * 4
    xs = [o.clone().to(DEV).requires_grad_(True) for o in outs]
    got = losses.multi_task_loss(xs, [t.to(DEV) for t in tgts], fns, w)
    xo = [o.clone().requires_grad_(True) for o in outs]
    want = cpu_ref.loss_list(xo, tgts, w, cad120=False)
    for a, b in zip(got, want):
        a, b = float(a.detach()), float(b.detach())
        assert (a != a and b != b) or abs(a - b) <= 1e-5 * max(1.0, abs(b))  # all-ignored NLL term: NaN on both sides
    sum(got).backward()
    sum(want).backward()
    for a, b in zip(xs, xo):
        gb = b.grad if b.grad is not None else torch.zeros_like(b)  # all-ignored BCE / budget term: constant 0 in torch
        ga = a.grad.cpu() if a.grad is not None else torch.zeros_like(b)
        assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-7)


def test_bench_size_properties():
    bs, C, T, E = 64, 13, 120, 2
    g = torch.Generator().manual_seed(0)
    x = torch.log_softmax(torch.randn(bs, C, T, E, generator=g), 1).to(DEV)
    t = torch.randint(0, C, (bs, T, E), generator=g).to(DEV)
    vals, grads = [], []
    for _ in range(2):
        xi = x.clone().requires_grad_(True)
        v = losses.multi_task_loss([xi, xi], [t, t], (losses.nll_loss,) * 2, [1.0, 0.25])
        (v[0] + v[1]).backward()
        vals.append(torch.stack(v).detach().clone())
        grads.append(xi.grad.clone())
    assert torch.equal(vals[0], vals[1]) and torch.equal(grads[0], grads[1])       # deterministic
    assert abs(float(grads[0].double().sum()) + 1.25) < 1e-6                        # sum of d/dx of mean NLL = -w
    ref = torch.nn.functional.nll_loss(x, t)
    assert abs(float(vals[0][0]) - float(ref)) < 1e-5 * abs(float(ref))
