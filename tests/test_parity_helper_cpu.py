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


PLANTED = [   # (configuration switches, layer, T): one (row / pair, unit) of the layer is put exactly on zero through its bias
    (dict(attention_style='v1'), 'objects_to_human_message_att_mlp.0', 3),                # concat score, frame level
    (dict(attention_style='v1'), 'objects_to_object_segment_message_att_mlp.0', 2),      # concat score, segment level
    (dict(attention_style='v4'), 'humans_to_object_message_att_mlp', 3),                  # bilinear score, frame level
    (dict(attention_style='v4'), 'objects_to_human_segment_message_att_mlp', 2),         # bilinear score, segment level
    (dict(message_granularity='v2'), 'objects_to_object_message_mlp.0', 3),               # receiver-specific message
    (dict(message_type='v1'), 'human_object_pairwise_relation_mlp.0', 3),                 # relational: pairwise g
    (dict(message_type='v1'), 'object_human_full_relation_mlp.0', 3),                     # relational: full f
    (dict(discrete_networks_num_layers=3), 'update_object_segment_mlp.0', 3),             # gate network, first hidden layer
    (dict(discrete_networks_num_layers=3), 'update_object_segment_mlp.2', 3),             # gate network, second hidden layer
    (dict(add_time_position=1, time_position_strategy='u'), 'time_position_mlp.0', 3),    # position feature
]


@pytest.mark.parametrize('extra,layer,T', PLANTED, ids=[f'{p[1]}-{i}' for i, p in enumerate(PLANTED)])
def test_every_relu_layer_of_the_general_forms_is_covered(extra, layer, T):
    """tests/relu_boundary.py recomputes the pre-activations of EVERY ReLU layer the path has -- also the per-pair layers of
    the general message forms, the attention-score functions, the hidden layers of the gate networks and the position
    features. For each: a unit planted on the boundary is found in the layer that owns it, and
    condition_case moves the case off it. (Segment-level scores are planted in a chain of two steps, whose first states
    do not depend on the score function: a softmax over equal scores.)"""
    from twog_gcn_amd.models import TGGCN
    torch.manual_seed(3)
    N, h, bs, H, O = 26, 16, 2, 2, 4
    cfg = dict(tp.STAGE1)
    cfg.update(extra)
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(13, None), hidden_size=h, gcn_node=N, **cfg).train()
    x_human, x_objects, mask = tp._synthetic(bs, T, H, O, N, 2)
    mask[0, -1] = 0
    m._gumbel_noise_override = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * (H + O), bs, 2))
    steps = torch.full((bs,), float(T))

    def fwd():
        return m(x_human, x_objects, mask, steps_per_example=steps)

    P = dict(m.named_parameters())
    for rnd in range(3):   # (the planted bias is a rounded fp32 number: a second pass removes what the rounding left)
        its = [i for i in relu_boundary._layers(m, fwd()) if i.layer == layer]
        assert its, ('layer not covered', layer, sorted({i.layer for i in relu_boundary._layers(m, fwd())}))
        pre, mag = its[0].pre_mag()
        row = int(mag[:, 0].argmax())                    # a row / pair that exists (and whose features are not all zero)
        with torch.no_grad():
            P[layer + '.bias'][0] -= pre[row, 0].to(torch.float32)
    found = relu_boundary.boundary_layers(m, fwd(), width=8.0)
    assert 0 in found.get(layer, torch.tensor([])).tolist(), (layer, found)
    rounds, nudged = relu_boundary.condition_case(m, fwd)
    assert rounds >= 1 and nudged.get(layer, 0) >= 1
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
