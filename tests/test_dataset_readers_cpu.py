"""CPU: the MPHOI-72 / Bimanual dataset readers (ground-truth JSON + zarr feature stores -> train / val / test loaders)
against golden G10 = the outputs of the REFERENCE's own readers (vhoi/data_loading.py:63-160,234-309) on the committed
dataset directories under tests/golden/g9_featstore/datasets (tools/make_golden_featstore.py)."""
import json
import os
import re

import numpy as np
import pytest

from tests.helpers import ROOT

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import data_loading as dl

DS = os.path.join(ROOT, 'tests', 'golden', 'g9_featstore', 'datasets')


@pytest.fixture(scope='module')
def g10():
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g10_loaders.npz')))


def _compare(g10, prefix, loaders):
    n = 0
    for tag, loader in loaders.items():
        tensors = loader.dataset.tensors
        assert len(tensors) == sum(1 for k in g10 if re.fullmatch(rf'{prefix}_{tag}_\d+', k))
        for i, t in enumerate(tensors):
            want = g10[f'{prefix}_{tag}_{i}']
            got = t.numpy()
            assert got.shape == want.shape and got.dtype == want.dtype, (prefix, tag, i, got.shape, want.shape)
            if got.dtype.kind == 'f':
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6, err_msg=f'{prefix} {tag} {i}')
            else:
                assert np.array_equal(got, want), (prefix, tag, i)
            n += 1
    return n


def test_mphoi_readers_match_the_reference(g10):
    d = os.path.join(DS, 'MPHOI')
    paths = [os.path.join(d, n) for n in ('mphoi_ground_truth_labels.json', 'faster_rcnn.zarr',
                                          'object_bounding_boxes.zarr', 'human_bounding_boxes.zarr', 'human_pose.zarr')]
    tr, va, info, scalers = dl.load_mphoi_training_data(*paths, '2G-GCN', 'multiple', test_subject_id='Subject14',
                                                        batch_size=2, val_fraction=0.4, seed=42,
                                                        scaling_strategy='standard', sigma=0.0, downsampling=2)
    te, info_t, seg, ids = dl.load_mphoi_testing_data(*paths, '2G-GCN', 'multiple', test_subject_id='Subject14',
                                                      batch_size=2, scalers=scalers, downsampling=2)
    assert _compare(g10, 'mphoi', {'train': tr, 'val': va, 'test': te}) >= 30
    assert tuple(info['input_size']) == tuple(g10['mphoi_input_size']) == tuple(info_t['input_size'])
    assert '|'.join(ids) == str(g10['mphoi_test_ids']) and seg is None
    for k, sc in scalers.items():
        np.testing.assert_allclose(sc.mean_, g10[f'mphoi_{k}_mean'], rtol=2e-5, atol=1e-6)  # float32 statistics
        np.testing.assert_allclose(sc.scale_, g10[f'mphoi_{k}_scale'], rtol=2e-5, atol=1e-6)
    assert len(tr.dataset) == 2 and len(va.dataset) == 2 and va.batch_size == 2


def test_bimanual_readers_match_the_reference_including_fps_doubling(g10):
    d = os.path.join(DS, 'BimanualActions')
    paths = [os.path.join(d, n) for n in ('bimacs_ground_truth_labels.json', 'faster_rcnn.zarr', 'bounding_boxes.zarr',
                                          'hands_pose.zarr')]
    fps = json.load(open(os.path.join(d, 'video_id_to_video_fps.json')))
    assert 15 in fps.values()
    tr, va, info, scalers = dl.load_bimanual_training_data(*paths, '2G-GCN', 'multiple', test_subject_id=1,
                                                           video_id_to_video_fps=dict(fps), batch_size=2,
                                                           val_fraction=0.25, seed=7, scaling_strategy=None, sigma=0.0,
                                                           downsampling=1)
    te, info_t, seg, ids = dl.load_bimanual_testing_data(*paths, '2G-GCN', 'multiple', test_subject_id=1,
                                                         video_id_to_video_fps=dict(fps), batch_size=2, scalers=scalers,
                                                         downsampling=1)
    assert _compare(g10, 'bimanual', {'train': tr, 'val': va, 'test': te}) >= 30
    assert tuple(info['input_size']) == tuple(g10['bimanual_input_size'])
    assert '|'.join(ids) == str(g10['bimanual_test_ids']) and scalers == {}


def test_split_is_the_reference_shuffle():
    import random
    a = list(range(11))
    train, test = dl.split_train_test(list(a), test_fraction=0.3, seed=5)
    random.seed(5)
    b = list(a)
    random.shuffle(b)
    assert test == b[:3] and train == b[3:]
