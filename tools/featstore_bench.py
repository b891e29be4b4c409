#!/usr/bin/env python3
"""Host-side throughput of the feature-store reader on a C3-sized clip (T=120, 2 humans x 2184 + 8 objects x 2048 fp32
features = 9.96 MB): chunk file -> pinned tensor -> HBM. Prints one JSON line.

The Blosc frames are made with the image's own c-blosc when it is there (the reference's default compressor settings);
without it only the uncompressed path is measured. Thread scaling is reported because the GPU boxes grant a CPU quota
(16 of 256 cores), not the core count the OS shows."""
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd import featstore  # noqa: E402
from twog_gcn_amd.hostcpu import effective_cpu_count, limit_host_threads  # noqa: E402

limit_host_threads()
rng = np.random.default_rng(0)
clip = {'Human1': (120, 2184), 'Human2': (120, 2184), 'objects': (120, 8, 2048)}
arrays = {k: np.maximum(rng.standard_normal(s), 0).astype(np.float32) for k, s in clip.items()}
nbytes = sum(a.nbytes for a in arrays.values())
out = {'clip_MB': nbytes / 1e6, 'cpus': effective_cpu_count()}


def best(fn, reps=7):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts)


with tempfile.TemporaryDirectory() as d:
    g = featstore.group(store=featstore.DirectoryStore(os.path.join(d, 'raw.zarr'))).create_group('v')
    for k, a in arrays.items():
        g.array(k, a, chunks=False, dtype=np.float32)
    node = featstore.open(os.path.join(d, 'raw.zarr'))['v']
    bufs = {k: torch.empty(s, dtype=torch.float32, pin_memory=torch.cuda.is_available()) for k, s in clip.items()}
    dt = best(lambda: [node[k].read_into(bufs[k]) for k in clip])
    out['raw_read_into_pinned_GBps'] = nbytes / dt / 1e9
    assert all(np.array_equal(bufs[k].numpy(), arrays[k]) for k in clip)

    blosc = '/opt/conda/lib/libblosc.so.1'
    if os.path.exists(blosc):
        lib = ctypes.CDLL(blosc)
        lib.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        lib.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        bdir = os.path.join(d, 'blosc.zarr', 'v')
        os.makedirs(bdir)
        for p in (os.path.join(d, 'blosc.zarr'), bdir):
            json.dump({'zarr_format': 2}, open(os.path.join(p, '.zgroup'), 'w'))
        frames, cbytes = {}, 0
        for k, a in arrays.items():
            dst = np.empty(a.nbytes + 16, np.uint8)
            n = lib.blosc_compress_ctx(5, 1, 4, a.nbytes, a.ctypes.data, dst.ctypes.data, dst.nbytes, b'lz4', 0, 1)
            frames[k] = dst[:n].copy()
            cbytes += n
            os.makedirs(os.path.join(bdir, k))
            frames[k].tofile(os.path.join(bdir, k, '.'.join('0' * a.ndim)))
            json.dump({'zarr_format': 2, 'shape': list(a.shape), 'chunks': list(a.shape), 'dtype': '<f4', 'order': 'C',
                       'compressor': {'id': 'blosc', 'cname': 'lz4', 'clevel': 5, 'shuffle': 1, 'blocksize': 0},
                       'fill_value': 0.0, 'filters': None}, open(os.path.join(bdir, k, '.zarray'), 'w'))
        out['blosc_ratio'] = cbytes / nbytes
        node = featstore.open(os.path.join(d, 'blosc.zarr'))['v']
        for nt in (1, 2, 4, 8, 16):
            dt = best(lambda: [node[k].read_into(bufs[k], n_threads=nt) for k in clip])
            out[f'blosc_read_into_pinned_GBps_t{nt}'] = nbytes / dt / 1e9
        assert all(np.array_equal(bufs[k].numpy(), arrays[k]) for k in clip)
        tmp = {k: np.empty(a.nbytes, np.uint8) for k, a in arrays.items()}
        for nt in (1, 8):
            dt = best(lambda: [lib.blosc_decompress_ctx(frames[k].ctypes.data, tmp[k].ctypes.data, tmp[k].nbytes, nt)
                               for k in clip])
            out[f'cblosc_decode_only_GBps_t{nt}'] = nbytes / dt / 1e9
    if torch.cuda.is_available():
        dev = {k: torch.empty(s, dtype=torch.float32, device='cuda:0') for k, s in clip.items()}
        torch.cuda.synchronize()

        def h2d():
            for k in clip:
                dev[k].copy_(bufs[k], non_blocking=True)
            torch.cuda.synchronize()
        out['pinned_h2d_GBps'] = nbytes / best(h2d) / 1e9
print(json.dumps(out))
