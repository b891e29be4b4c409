"""GPU: feature stores -> pinned staging -> HBM. The decoded arrays land in pinned host tensors (no pageable
temporary), the asynchronous H2D copy delivers them bit-exact, and a model step consumes a batch assembled from a store."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import ROOT

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import featstore

pytestmark = pytest.mark.gpu
STORES = os.path.join(ROOT, 'tests', 'golden', 'g9_featstore', 'stores')


def test_store_to_pinned_to_device_is_bit_exact():
    expected = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g9_featstore', 'expected.npz')))
    root = featstore.open(os.path.join(STORES, 'features.zarr'))
    stream = torch.cuda.Stream()
    for vid in root:
        names = list(root[vid].array_keys())
        host = featstore.load_pinned(root[vid], names)
        with torch.cuda.stream(stream):
            dev = {n: t.to('cuda:0', non_blocking=True) for n, t in host.items()}
        stream.synchronize()
        for n in names:
            assert host[n].is_pinned()
            want = expected[f'features.zarr|{vid}/{n}']
            assert dev[n].dtype == torch.float32 and tuple(dev[n].shape) == want.shape
            assert np.array_equal(dev[n].cpu().numpy(), want), (vid, n)


def test_large_store_round_trip_through_the_device(tmp_path):
    """Full-size clip (T=120, 2048 features, 8 objects): write -> read into pinned memory -> HBM -> checksum on device."""
    rng = np.random.default_rng(0)
    g = featstore.group(store=featstore.DirectoryStore(str(tmp_path / 'big.zarr')))
    v = g.create_group('video')
    obj = np.maximum(rng.standard_normal((120, 8, 2048)), 0).astype(np.float32)
    v.array('objects', obj, chunks=False, dtype=np.float32)
    host = featstore.load_pinned(featstore.open(str(tmp_path / 'big.zarr'))['video'], ['objects'])['objects']
    dev = host.to('cuda:0', non_blocking=True)
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu(), torch.from_numpy(obj))
    assert abs(dev.double().sum().item() - obj.astype(np.float64).sum()) < 1e-6 * obj.size
