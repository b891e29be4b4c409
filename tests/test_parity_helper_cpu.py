"""CPU: the full-path comparison protocol of tests/test_parity_gpu.py (_oracle_vs_hip: ReLU-boundary conditioning of the
case, 5e-4 hard gate, fp64 yardstick) exercised on the torch test double of the kernel interface, so that the protocol
itself is covered without a GPU -- including the CAD-120 layout (12 outputs, both segmentations given) and the
"last two objects virtual on half the clips" input variant."""
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import kernels as twog_kernels
from tests.fake_kernels import FakeKernels
import tests.test_parity_gpu as tp
from tests import relu_boundary


@pytest.fixture(autouse=True)
def fake_backend(monkeypatch):
    twog_kernels._set_backend_for_tests(FakeKernels())
    monkeypatch.setattr(tp, 'DEV', 'cpu')
    yield
    twog_kernels._set_backend_for_tests(None)


def test_protocol_c2_layout():
    tp._oracle_vs_hip(bs=2, T=5, H=2, O=4, N=26, h=32, backward=True, seed=5)


def test_protocol_c1_layout_both_segmentations_given():
    tp._oracle_vs_hip(bs=2, T=4, H=1, O=5, N=19, h=32, backward=True, seed=21, n_sub=10, n_aff=12, both_given=True)


def test_protocol_virtual_objects_on_half_the_clips():
    tp._oracle_vs_hip(bs=4, T=3, H=2, O=8, N=34, h=32, backward=True, seed=23, virtual='half')


def test_conditioning_moves_a_case_off_the_relu_boundary():
    """A unit forced onto the boundary (bias set so that one row's pre-activation is ~0) is found and nudged away; the
    case then reports no boundary unit at the conditioning width."""
    from twog_gcn_amd.models import TGGCN
    torch.manual_seed(1)
    N, h, bs, T, H, O = 26, 16, 2, 3, 2, 4
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(13, None), hidden_size=h, gcn_node=N, **tp.STAGE1).train()
    x_human, x_objects, mask = tp._synthetic(bs, T, H, O, N, 2)
    seg = torch.ones(bs, T, H)
    m._gumbel_noise_override = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, bs, 2))

    def fwd():
        return m(x_human, x_objects, mask, human_segmentation=seg)

    with torch.no_grad():   # put row 0 of unit 3 of the object embedding exactly on zero
        lin = m.object_embedding_mlp[0]
        pre = (x_objects.view(-1, 2048)[0].double() @ lin.weight[3].double()).item()
        lin.bias[3] = -pre
    assert 3 in relu_boundary.boundary_layers(m, fwd(), width=8.0).get('object_embedding_mlp.0', torch.tensor([])).tolist()
    rounds, nudged = relu_boundary.condition_case(m, fwd)
    assert rounds >= 1 and nudged.get('object_embedding_mlp.0', 0) >= 1
    assert not relu_boundary.boundary_layers(m, fwd(), width=8.0)


def test_a_planted_one_percent_gradient_error_fails_the_gate(monkeypatch):
    """The gate must catch what the old escape clause let through: a 1 % error in ONE kernel's backward (here the ReLU
    backward of the test double scaled by 1.01) moves the gradients upstream of it by about a percent -- far beyond
    5e-4, with no boundary unit to blame -- and the comparison has to fail."""
    orig = FakeKernels.relu_bwd

    def off_by_one_percent(self, dy, y, dx=None):
        return orig(self, dy * 1.01, y, dx)

    monkeypatch.setattr(FakeKernels, 'relu_bwd', off_by_one_percent)
    with pytest.raises(AssertionError, match='gradients beyond 5e-4|beyond the fp64 yardstick'):
        tp._oracle_vs_hip(bs=2, T=5, H=2, O=4, N=26, h=32, backward=True, seed=5)
