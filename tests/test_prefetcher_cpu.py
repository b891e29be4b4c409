"""CPU: DevicePrefetcher (host logic of the input pipeline) yields exactly what the reference's
loop `for batch in loader: fetch(batch, device)` yields, in order, for ragged last batches and shuffled loaders."""
import torch
from torch.utils.data import DataLoader, TensorDataset

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd.data_loading import DevicePrefetcher, gcn_fetcher


def make_loader(n, bs, shuffle, seed=0):
    g = torch.Generator().manual_seed(seed)
    tensors = [torch.randn(n, 5, 2, 7, generator=g), torch.randn(n, 5, 3, 4, generator=g), torch.ones(n, 3),
               torch.ones(n, 5, 2), torch.randn(n, 5, 2, 2, generator=g), torch.randn(n, 5, 2, 3, generator=g), torch.randn(n, 5, 3, 3, generator=g),
               torch.full((n,), 5.0), torch.randint(0, 4, (n, 5, 2), generator=g), torch.randint(0, 4, (n, 5, 2), generator=g)]
    gen = torch.Generator().manual_seed(seed + 1)
    return DataLoader(TensorDataset(*tensors), batch_size=bs, shuffle=shuffle, generator=gen if shuffle else None)


def flatten(batches):
    return [[t for group in b for t in group] for b in batches]


def test_same_batches_as_the_plain_loop():
    kw = dict(dataset_name='mphoi', input_human_segmentation=True)
    for n, bs in ((7, 3), (4, 4), (1, 2), (9, 2)):
        loader = make_loader(n, bs, shuffle=False)
        want = flatten([gcn_fetcher(b, device='cpu', **kw) for b in loader])
        got = flatten(list(DevicePrefetcher(loader, gcn_fetcher, 'cpu', **kw)))
        assert len(got) == len(want) == len(loader)
        for gb, wb in zip(got, want):
            assert all(torch.equal(a, b) for a, b in zip(gb, wb))


def test_shuffled_loader_order_is_the_loaders():
    kw = dict(dataset_name='mphoi')
    a, b = make_loader(10, 4, shuffle=True, seed=3), make_loader(10, 4, shuffle=True, seed=3)
    want = flatten([gcn_fetcher(x, device='cpu', **kw) for x in a])
    got = flatten(list(DevicePrefetcher(b, gcn_fetcher, 'cpu', **kw)))
    for gb, wb in zip(got, want):
        assert all(torch.equal(x, y) for x, y in zip(gb, wb))


def test_resident_mode_on_cpu_device():
    kw = dict(dataset_name='mphoi')
    loader = make_loader(7, 3, shuffle=False)
    want = flatten([gcn_fetcher(x, device='cpu', **kw) for x in loader])
    got = flatten(list(DevicePrefetcher(loader, gcn_fetcher, 'cpu', resident=True, **kw)))
    assert len(got) == len(want)
    for gb, wb in zip(got, want):
        assert all(torch.equal(x, y) for x, y in zip(gb, wb))
