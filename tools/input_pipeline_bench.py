#!/usr/bin/env python3
"""Host -> HBM input pipeline rates for the C3 batch shape (bs64, T=120: 637 MB per batch), SURVEY section 8f row 2:
(a) the reference's loop: DataLoader collate + synchronous pageable .to(device) per slot (vhoi/data_loading.py:376,
:1284-1314); (b) DevicePrefetcher: gather into pinned slots + non-blocking copies on a side stream; (c) HBM-resident
split. Prints clips/s of the pipeline alone and, for (b)/(c), under a busy compute stream."""
import os
import sys
import time
import torch
from torch.utils.data import DataLoader, TensorDataset
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.data_loading import DevicePrefetcher, gcn_fetcher

dev = 'cuda:0'
n, bs, T, H, O, N = 256, 64, 120, 2, 8, 34
g = torch.Generator().manual_seed(0)
tensors = [torch.randn(n, T, H, 2048 + 4 * N, generator=g), torch.randn(n, T, O, 2048, generator=g), torch.ones(n, O),
           torch.ones(n, T, H), torch.zeros(n, T, H, H), torch.zeros(n, T, H, O), torch.zeros(n, T, O, O),
           torch.full((n,), float(T)), torch.randint(0, 13, (n, T, H)), torch.randint(0, 13, (n, T, H))]
loader = DataLoader(TensorDataset(*tensors), batch_size=bs, shuffle=False)
kw = dict(dataset_name='mphoi')
mb = sum(t[0].numel() * t.element_size() for t in tensors[:3]) * bs / 1e6


def rate(it, epochs=2, busy=False):
    a = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    clips = 0
    for _ in range(epochs):
        for data, target in it():
            clips += data[0].shape[0]
            if busy:
                for _ in range(40):
                    a = a @ a * 1e-4
    torch.cuda.synchronize()
    return clips / (time.perf_counter() - t0)


ref = lambda: (gcn_fetcher(b, device=dev, **kw) for b in loader)
pre = DevicePrefetcher(loader, gcn_fetcher, dev, **kw)
res = DevicePrefetcher(loader, gcn_fetcher, dev, resident=True, **kw)
rate(ref, 1); rate(lambda: iter(pre), 1)
print(f'batch = {mb:.0f} MB of moved tensors, {n} clips in the split, host threads {torch.get_num_threads()}')
print(f'(a) reference-style synchronous pageable copies : {rate(ref):8.0f} clips/s')
print(f'(b) DevicePrefetcher pinned double buffer        : {rate(lambda: iter(pre)):8.0f} clips/s')
print(f'(c) DevicePrefetcher resident split              : {rate(lambda: iter(res)):8.0f} clips/s')
print(f'(a) with a busy compute stream                   : {rate(ref, busy=True):8.0f} clips/s')
print(f'(b) with a busy compute stream                   : {rate(lambda: iter(pre), busy=True):8.0f} clips/s')
print(f'(c) with a busy compute stream                   : {rate(lambda: iter(res), busy=True):8.0f} clips/s')
