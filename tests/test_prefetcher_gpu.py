"""GPU: the double-buffered pinned input pipeline and the HBM-resident mode deliver bit-identical batches to the
reference-style synchronous loop, also while a compute kernel stream is busy (copy/compute overlap on separate streams)."""
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd.data_loading import DevicePrefetcher, gcn_fetcher
from tests.test_prefetcher_cpu import make_loader, flatten

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('resident', [False, True])
@pytest.mark.parametrize('shuffle', [False, True])
def test_batches_equal_synchronous_loop(resident, shuffle):
    kw = dict(dataset_name='mphoi', input_human_segmentation=True)
    a, b = make_loader(11, 4, shuffle, seed=5), make_loader(11, 4, shuffle, seed=5)
    want = flatten([gcn_fetcher(x, device=DEV, **kw) for x in a])
    got = []
    busy = torch.randn(2048, 2048, device=DEV)
    for batch in DevicePrefetcher(b, gcn_fetcher, DEV, resident=resident, **kw):
        busy = busy @ busy * 1e-3          # keep the compute stream busy between hand-overs
        got.append([t.clone() for group in batch for t in group])
    torch.cuda.synchronize()
    assert len(got) == len(want) == 3
    for gb, wb in zip(got, want):
        for x, y in zip(gb, wb):
            # resident mode keeps every slot in HBM, also those the fetcher would leave on the host (unused by the feeder)
            assert (resident or x.device == y.device) and torch.equal(x.cpu(), y.cpu())


def test_pinned_slots_are_reused_not_regrown():
    kw = dict(dataset_name='mphoi')
    p = DevicePrefetcher(make_loader(16, 4, False), gcn_fetcher, DEV, **kw)
    list(p)
    ptrs = [{i: b.data_ptr() for i, b in s.items()} for s in p._slots]
    list(p)
    assert ptrs == [{i: b.data_ptr() for i, b in s.items()} for s in p._slots]
    assert all(b.is_pinned() for s in p._slots for b in s.values())
