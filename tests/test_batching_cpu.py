"""CPU: the batching mirror (2g-gcn_amd/data_loading.py) against tensors produced by the reference's own
create_data_loader / gcn_fetcher / gcn_forward on synthetic raw videos (fixture G6, tools/make_golden.py)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import data_loading as dl
from tests.helpers import GOLDEN, ROOT


class _Seg:  # attribute bag standing in for the reference's CAD120VideoSegment
    def __init__(self):
        self.start_frame = self.end_frame = self.subactivity = self.next_subactivity = None
        self.object_affordance, self.next_object_affordance = {}, {}


def raw_videos(kind, seed):
    """Same generator as tools/make_golden.py::_raw_videos (kept in sync by the fixture comparison itself)."""
    rng = np.random.RandomState(seed)
    vids, F = [], 16
    if kind in ('mphoi', 'bimanual'):
        J = 32 if kind == 'mphoi' else 21
        n_obj_max = 4 if kind == 'mphoi' else 9
        keys = ('Human1', 'Human2') if kind == 'mphoi' else ('left_hand', 'right_hand')
        for L, n in ((20, n_obj_max), (14, n_obj_max - 1), (17, 2)):
            gt = {}
            for k in keys:
                y = []
                while len(y) < L:
                    y += [int(rng.randint(0, 5))] * int(rng.randint(2, 6))
                gt[k] = y[:L]
            vids.append([rng.randn(L, F).astype(np.float32), rng.randn(L, F).astype(np.float32),
                         rng.randn(L, n, F).astype(np.float32), gt,
                         rng.rand(L, 4) * 1000, rng.rand(L, 4) * 1000, rng.rand(L, n, 4) * 1000,
                         rng.rand(L, J, 2) * 1000, rng.rand(L, J, 2) * 1000])
    else:
        for L, n in ((21, 5), (15, 3), (18, 4)):
            segs, start = [], 1
            while start <= L:
                end = min(L, start + int(rng.randint(2, 6)))
                s = _Seg()
                s.start_frame, s.end_frame = start, end
                s.subactivity = int(rng.randint(1, 11))
                s.object_affordance = {o + 1: int(rng.randint(1, 13)) for o in range(n)}
                segs.append(s)
                start = end + 1
            for a, b in zip(segs[:-1], segs[1:]):
                a.next_subactivity = b.subactivity
                a.next_object_affordance = dict(b.object_affordance)
            vids.append([rng.randn(L, F).astype(np.float32), rng.randn(L, n, F).astype(np.float32),
                         rng.rand(L, 4) * 400, rng.rand(L, n, 4) * 400, rng.rand(L, 9, 2) * 300, segs])
    return vids


@pytest.mark.parametrize('kind', ['mphoi', 'bimanual', 'cad120'])
@pytest.mark.parametrize('sigma,test_data', [(0.0, False), (2.0, False), (0.0, True)])
def test_create_data_loader_matches_reference(kind, sigma, test_data):
    z = np.load(f'{GOLDEN}/g6_batching.npz')
    loader, scalers, segs = dl.create_data_loader(raw_videos(kind, 60), '2G-GCN', 'multiple', kind, batch_size=2,
                                                  shuffle=False, sigma=sigma, downsampling=3, test_data=test_data)
    tensors = loader.dataset.tensors
    keys = [k for k in z.files if k.startswith(f'{kind}_s{sigma}_t{int(test_data)}_')]
    assert len(tensors) == len(keys) == (20 if kind == 'cad120' else 14)
    for i, t in enumerate(tensors):
        ref = z[f'{kind}_s{sigma}_t{int(test_data)}_{i}']
        assert tuple(t.shape) == ref.shape, (i, t.shape, ref.shape)
        assert str(t.numpy().dtype) == str(ref.dtype), (i, t.dtype, ref.dtype)
        if ref.dtype.kind == 'i':
            assert np.array_equal(t.numpy(), ref), i
        else:
            assert np.allclose(t.numpy(), ref, rtol=1e-6, atol=1e-6), (i, np.abs(t.numpy() - ref).max())
    assert not any(torch.isnan(t).any() for t in tensors if t.is_floating_point())
    assert dl.input_size_from_data_loader(loader, '2G-GCN', 'multiple') == (16 + tensors[0].shape[-1] - 16, 16)
    if kind == 'cad120':
        assert len(segs) == 3 and all(isinstance(s, list) for s in segs)


@pytest.mark.parametrize('kind', ['mphoi', 'bimanual', 'cad120'])
def test_fetcher_and_feeder_match_reference(kind):
    z = np.load(f'{GOLDEN}/g6_batching.npz')
    loader, _, _ = dl.create_data_loader(raw_videos(kind, 60), '2G-GCN', 'multiple', kind, batch_size=2, shuffle=False,
                                         downsampling=3)
    batch = next(iter(loader))

    class Rec:
        def __call__(self, **kw):
            self.kw = kw
            return 'out'

    for tag, kwargs in (('plain', dict(dataset_name=kind, impose_segmentation_pattern=1)),
                        ('input', dict(dataset_name=kind, input_human_segmentation=True, input_object_segmentation=True,
                                       make_attention_distance_based=True))):
        fetch = dl.select_model_data_fetcher('2G-GCN', 'multiple', **kwargs)
        feed = dl.select_model_data_feeder('2G-GCN', 'multiple', **kwargs)
        data, targets = fetch(batch, device='cpu')
        rec = Rec()
        assert feed(rec, data) == 'out'
        assert len(targets) == int(z[f'{kind}_{tag}_n_targets'])
        for k, v in rec.kw.items():
            if torch.is_tensor(v):
                assert np.allclose(v.numpy(), z[f'{kind}_{tag}_kw_{k}'], rtol=1e-6, atol=1e-6), (tag, k)
            else:
                assert f'{kind}_{tag}_kwnone_{k}' in z.files, (tag, k)
    with pytest.raises(ValueError):
        dl.gcn_forward(Rec(), data, impose_segmentation_pattern=2)
    with pytest.raises(KeyError):
        dl.select_model_data_fetcher('cad120_baseline', 'multiple')
    assert dl.determine_num_classes('2G-GCN', 'multiple', 'mphoi') == (13, None)
    assert dl.determine_num_classes('2G-GCN', 'multiple', 'bimanual') == (14, None)
    assert dl.determine_num_classes('2G-GCN', 'multiple', 'cad120') == (10, 12)


def test_segmentation_helpers_bit_exact_with_reference():
    """Golden G11 (tools/make_golden.py::g11_segmentation_helpers, outputs of the reference's own helpers): end flags from
    labels, the Gaussian budget targets -- bit for bit, including what the call leaves in the caller's array -- and the
    cleared last end flag."""
    z = np.load(os.path.join(GOLDEN, 'g11_segmentation_helpers.npz'))
    for ci in range(3):
        labels = z[f'c{ci}_labels']
        for style in ('input', 'output'):
            got = dl.segmentation_from_output_class(labels.copy(), segmentation_type=style)
            assert np.array_equal(np.asarray(got, dtype=np.float32), z[f'c{ci}_seg_{style}'].astype(np.float32)), (ci, style)
        seg = z[f'c{ci}_seg_output'].astype(np.float32)
        for sigma in (0.0, 1.0, 2.0, 3.5):
            arg = seg.copy()
            res = dl.smooth_segmentation(arg, sigma)
            want = z[f'c{ci}_smooth_{sigma}']
            assert res.dtype == want.dtype and np.array_equal(res, want), (ci, sigma, float(np.abs(res - want).max()))
            assert np.array_equal(arg, z[f'c{ci}_smooth_{sigma}_arg_after']), (ci, sigma, 'caller array')
        got = dl.ignore_last_step_end_flag(z[f'c{ci}_seg_input'].astype(np.float32).copy())
        assert np.array_equal(got, z[f'c{ci}_ignore_last'])
