#!/usr/bin/env python3
"""Generates golden G9 (tests/golden/g9_featstore): feature stores in the reference's on-disk format.

Build container only. zarr / numcodecs are not installable here, but the REAL c-blosc (libblosc 1.21.0) ships in the
image's conda tree; it is driven through ctypes with exactly the call numcodecs.Blosc.encode makes for zarr's default
compressor (numcodecs/blosc.pyx: blosc_compress_ctx(clevel, shuffle, itemsize, nbytes, src, dest, destsize, cname,
blocksize, 1) with cname='lz4', clevel=5, shuffle=SHUFFLE(1), blocksize=0), so every chunk file below is a genuine
Blosc frame. The directory layout (.zgroup / .zarray / chunk key '0.0') is written by hand after the zarr v2 storage
spec, mirroring what `group.array(name, data, chunks=False, dtype=np.float32)` (vhoi/roi_features.py:227-242) produces.

Outputs (all small):
  stores/features.zarr/<video>/<name>       Blosc-compressed single-chunk fp32 arrays (the reference's case)
  stores/variants.zarr/...                  uncompressed, chunk-grid, F-order, zlib, big-endian variants
  expected.npz                              the arrays, keyed '<store>|<path>'
  frames.npz                                stand-alone Blosc frames (many codec / flag combinations) + their payloads
"""
import ctypes
import json
import os
import shutil
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden', 'g9_featstore')
BLOSC = '/opt/conda/lib/libblosc.so.1.21.0'

lib = ctypes.CDLL(BLOSC)
lib.blosc_get_version_string.restype = ctypes.c_char_p
lib.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
lib.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def blosc_encode(buf: bytes, typesize: int, cname=b'lz4', clevel=5, shuffle=1, blocksize=0) -> bytes:
    src = np.frombuffer(buf, np.uint8)
    dst = np.empty(src.nbytes + 16, np.uint8)
    n = lib.blosc_compress_ctx(clevel, shuffle, typesize, src.nbytes, src.ctypes.data, dst.ctypes.data, dst.nbytes,
                               cname, blocksize, 1)
    assert n > 0, n
    frame = dst[:n].tobytes()
    back = np.empty(src.nbytes, np.uint8)   # the real library round-trips its own frame
    assert lib.blosc_decompress_ctx(np.frombuffer(frame, np.uint8).ctypes.data, back.ctypes.data, back.nbytes, 1) == src.nbytes
    assert back.tobytes() == buf
    return frame


def write_json(path, doc):
    with open(path, 'w') as f:
        json.dump(doc, f, indent=4, sort_keys=True)


def make_group(path):
    os.makedirs(path, exist_ok=True)
    write_json(os.path.join(path, '.zgroup'), {'zarr_format': 2})


BLOSC_DEFAULT = {'id': 'blosc', 'cname': 'lz4', 'clevel': 5, 'shuffle': 1, 'blocksize': 0}


def write_array(path, a, compressor=BLOSC_DEFAULT, chunks=None, order='C', fill_value=0.0, skip=()):
    os.makedirs(path, exist_ok=True)
    chunks = tuple(chunks or a.shape)
    write_json(os.path.join(path, '.zarray'), {
        'zarr_format': 2, 'shape': list(a.shape), 'chunks': list(chunks), 'dtype': a.dtype.str,
        'compressor': compressor, 'fill_value': fill_value, 'order': order, 'filters': None})
    grid = [-(-s // c) for s, c in zip(a.shape, chunks)]
    for idx in np.ndindex(*grid):
        if idx in skip:
            continue
        chunk = np.full(chunks, fill_value, a.dtype)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, a.shape))
        chunk[tuple(slice(0, s.stop - s.start) for s in sel)] = a[sel]
        raw = chunk.tobytes(order=order)
        if compressor is None:
            enc = raw
        elif compressor['id'] == 'blosc':
            enc = blosc_encode(raw, a.dtype.itemsize, compressor['cname'].encode(), compressor['clevel'],
                               compressor['shuffle'], compressor['blocksize'])
        elif compressor['id'] == 'zlib':
            enc = zlib.compress(raw, compressor['level'])
        with open(os.path.join(path, '.'.join(map(str, idx))), 'wb') as f:
            f.write(enc)


def _labels(rng, L):
    y = []
    while len(y) < L:
        y += [int(rng.integers(0, 5))] * int(rng.integers(2, 6))
    return y[:L]


def make_dataset_stores_and_loader_golden(rng):
    """Golden G10: MPHOI-72- and Bimanual-style dataset directories (ground-truth JSON + the feature / box / pose stores,
    Blosc chunks from the real c-blosc) and what the REFERENCE's own readers make of them. zarr is absent, so the
    reference's `import zarr` is satisfied with this repo's stand-in (featstore, itself pinned by G9): the reference
    code that runs here is its video filtering, fps doubling, train/val split and create_data_loader."""
    import sys
    import types
    REF = '/root/reference'
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    import torch  # noqa: F401
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import featstore
    sys.modules['zarr'] = featstore
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = object
    sys.modules.setdefault('torch.utils.tensorboard', tb)
    import vhoi.data_loading as ref_dl

    F = 16
    out = {}
    # ---- MPHOI-72 layout (conf/data/mphoi.yaml:3-7) ----
    d = os.path.join(OUT, 'datasets', 'MPHOI')
    names = ('faster_rcnn.zarr', 'object_bounding_boxes.zarr', 'human_bounding_boxes.zarr', 'human_pose.zarr')
    for n in names:
        make_group(os.path.join(d, n))
    gt = {}
    vids = [('Subject14-Cheering-1', 20, 4), ('Subject14-Co_working-2', 14, 3), ('Subject25-Cheering-1', 17, 2),
            ('Subject23-Hair_cutting-1', 15, 4), ('Subject45-Cheering-3', 12, 1), ('Subject25-Hair_cutting-2', 16, 2),
            ('Subject35-Co_working-1', 13, 3)]
    for vid, L, n_obj in vids:
        for n in names:
            make_group(os.path.join(d, n, vid))
        gt[vid] = {'Human1': _labels(rng, L), 'Human2': _labels(rng, L)}
        for h in ('Human1', 'Human2'):
            write_array(os.path.join(d, names[0], vid, h), np.maximum(rng.standard_normal((L, F)), 0).astype(np.float32))
            write_array(os.path.join(d, names[2], vid, h), (rng.random((L, 4)) * 1000).astype(np.float32))
            write_array(os.path.join(d, names[3], vid, h), (rng.random((L, 32, 2)) * 1000).astype(np.float32))
        write_array(os.path.join(d, names[0], vid, 'objects'), np.maximum(rng.standard_normal((L, n_obj, F)), 0).astype(np.float32))
        write_array(os.path.join(d, names[1], vid, 'objects'), (rng.random((L, n_obj, 4)) * 1000).astype(np.float32))
    write_json(os.path.join(d, 'mphoi_ground_truth_labels.json'), gt)
    paths = [os.path.join(d, 'mphoi_ground_truth_labels.json')] + [os.path.join(d, n) for n in
                                                                   (names[0], names[1], names[2], names[3])]
    tr, va, info, scalers = ref_dl.load_mphoi_training_data(*paths, '2G-GCN', 'multiple', test_subject_id='Subject14',
                                                            batch_size=2, val_fraction=0.4, seed=42,
                                                            scaling_strategy='standard', sigma=0.0, downsampling=2)
    te, info_t, seg, ids = ref_dl.load_mphoi_testing_data(*paths, '2G-GCN', 'multiple', test_subject_id='Subject14',
                                                          batch_size=2, scalers=scalers, downsampling=2)
    for tag, loader in (('train', tr), ('val', va), ('test', te)):
        for i, t in enumerate(loader.dataset.tensors):
            out[f'mphoi_{tag}_{i}'] = t.numpy()
    out['mphoi_input_size'] = np.array(info['input_size'])
    out['mphoi_test_ids'] = np.array('|'.join(ids))
    for k, sc in scalers.items():
        out[f'mphoi_{k}_mean'], out[f'mphoi_{k}_scale'] = sc.mean_, sc.scale_
    assert seg is None and len(tr.dataset) + len(va.dataset) == 4 and len(te.dataset) == 2, (len(tr.dataset), len(va.dataset))

    # ---- Bimanual Actions layout (conf/data/bimanual.yaml:3-6) ----
    d = os.path.join(OUT, 'datasets', 'BimanualActions')
    names = ('faster_rcnn.zarr', 'bounding_boxes.zarr', 'hands_pose.zarr')
    for n in names:
        make_group(os.path.join(d, n))
    gt, fps = {}, {}
    vids = [('subject_1-task_1_k_cooking-take_0', 18, 3, 30), ('subject_1-task_2_k_cooking_with_bowls-take_1', 9, 5, 15),
            ('subject_2-task_1_k_cooking-take_0', 16, 2, 30), ('subject_3-task_4_k_wiping-take_2', 8, 4, 15),
            ('subject_4-task_1_k_cooking-take_3', 15, 9, 30), ('subject_5-task_5_k_cereals-take_0', 14, 1, 30)]
    for vid, L, n_obj, f in vids:
        for n in names:
            make_group(os.path.join(d, n, vid))
        gt[vid] = {'left_hand': _labels(rng, L), 'right_hand': _labels(rng, L)}
        fps[vid] = f
        for h in ('left_hand', 'right_hand'):
            write_array(os.path.join(d, names[0], vid, h), np.maximum(rng.standard_normal((L, F)), 0).astype(np.float32))
            write_array(os.path.join(d, names[1], vid, h), (rng.random((L, 4)) * 480).astype(np.float32))
            write_array(os.path.join(d, names[2], vid, h), (rng.random((L, 21, 2)) * 480).astype(np.float32))
        write_array(os.path.join(d, names[0], vid, 'objects'), np.maximum(rng.standard_normal((L, n_obj, F)), 0).astype(np.float32))
        write_array(os.path.join(d, names[1], vid, 'objects'), (rng.random((L, n_obj, 4)) * 480).astype(np.float32))
    write_json(os.path.join(d, 'bimacs_ground_truth_labels.json'), gt)
    write_json(os.path.join(d, 'video_id_to_video_fps.json'), fps)
    paths = [os.path.join(d, 'bimacs_ground_truth_labels.json')] + [os.path.join(d, n) for n in names]
    tr, va, info, scalers = ref_dl.load_bimanual_training_data(*paths, '2G-GCN', 'multiple', test_subject_id=1,
                                                               video_id_to_video_fps=dict(fps), batch_size=2,
                                                               val_fraction=0.25, seed=7, scaling_strategy=None,
                                                               sigma=0.0, downsampling=1)
    te, info_t, seg, ids = ref_dl.load_bimanual_testing_data(*paths, '2G-GCN', 'multiple', test_subject_id=1,
                                                             video_id_to_video_fps=dict(fps), batch_size=2,
                                                             scalers=scalers, downsampling=1)
    for tag, loader in (('train', tr), ('val', va), ('test', te)):
        for i, t in enumerate(loader.dataset.tensors):
            out[f'bimanual_{tag}_{i}'] = t.numpy()
    out['bimanual_input_size'] = np.array(info['input_size'])
    out['bimanual_test_ids'] = np.array('|'.join(ids))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'g10_loaders.npz'), **out)
    print('g10:', len(out), 'arrays; train/val/test sizes', len(tr.dataset), len(va.dataset), len(te.dataset))


def main():
    print('c-blosc', lib.blosc_get_version_string().decode())
    if os.path.exists(OUT):
        shutil.rmtree(OUT)
    rng = np.random.default_rng(9)
    expected = {}

    # (1) the reference's layout: <store>/<video_id>/<entity> written with chunks=False and the default compressor
    feats = os.path.join(OUT, 'stores', 'features.zarr')
    make_group(feats)
    T, F, O = 10, 256, 3
    for vid in ('Subject14-Cheering-1', 'Subject25-Co_working-3'):
        make_group(os.path.join(feats, vid))
        arrays = {
            'Human1': np.maximum(rng.standard_normal((T, F)), 0).astype(np.float32),       # post-ReLU-like features
            'Human2': np.maximum(rng.standard_normal((T, F)), 0).astype(np.float32),
            'objects': np.maximum(rng.standard_normal((T, O, F)), 0).astype(np.float32),
            'Human1_bbs': (rng.random((T, 4)) * 640).round(1).astype(np.float32),
            'objects_bbs': (rng.random((T, O, 4)) * 640).round(1).astype(np.float32),
            'Human1_pose': (rng.random((T, 9, 2)) * 480).round(0).astype(np.float32),
        }
        for name, a in arrays.items():
            write_array(os.path.join(feats, vid, name), a)
            expected[f'features.zarr|{vid}/{name}'] = a
    write_json(os.path.join(feats, 'Subject14-Cheering-1', '.zattrs'), {'fps': 30, 'note': 'golden'})

    # (2) variants the format allows
    var = os.path.join(OUT, 'stores', 'variants.zarr')
    make_group(var)
    make_group(os.path.join(var, 'g'))
    base = rng.integers(0, 7, (13, 50)).astype(np.float32)
    cases = {
        'raw': dict(a=base, compressor=None),
        'g/grid': dict(a=base, chunks=(4, 16)),
        'g/grid_raw_missing': dict(a=None, compressor=None, chunks=(4, 16), skip=((0, 0),)),
        'forder': dict(a=base, order='F', chunks=(5, 20)),
        'zlib': dict(a=base, compressor={'id': 'zlib', 'level': 1}),
        'big_endian': dict(a=base.astype('>f4')),
        'f8': dict(a=rng.standard_normal((6, 3, 5))),
        'i8_lz4hc': dict(a=rng.integers(0, 100, (40, 9)), compressor=dict(BLOSC_DEFAULT, cname='lz4hc', clevel=9)),
        'noshuffle': dict(a=base, compressor=dict(BLOSC_DEFAULT, shuffle=0)),
        'blosc_zlib': dict(a=base, compressor=dict(BLOSC_DEFAULT, cname='zlib')),
        'vector': dict(a=np.arange(1000, dtype=np.float32)),
    }
    # grid_raw_missing: chunk (0, 0) is absent on disk -> reads as fill_value (0.0); make the expectation say so
    gm = base.copy()
    gm[:4, :16] = 0
    cases['g/grid_raw_missing']['a'] = gm
    for key, kw in cases.items():
        a = kw.pop('a')
        write_array(os.path.join(var, *key.split('/')), a, **kw)
        expected[f'variants.zarr|{key}'] = a
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **expected)

    # (3) stand-alone frames: flag / codec / size combinations of the container format
    frames = {}

    def add(name, a, **kw):
        a = np.ascontiguousarray(a)
        ts = kw.pop('typesize', a.dtype.itemsize)
        frames[f'frame_{name}'] = np.frombuffer(blosc_encode(a.tobytes(), ts, **kw), np.uint8)
        frames[f'data_{name}'] = np.frombuffer(a.tobytes(), np.uint8)

    relu = np.maximum(rng.standard_normal((24, 512)), 0).astype(np.float32)
    add('default_f4', relu)                                         # split streams, shuffle, one block
    add('tiny_memcpyed', np.arange(7, dtype=np.float32))            # < 128 bytes: stored, flags 0x2
    add('incompressible_u1', rng.integers(0, 256, 5000).astype(np.uint8))
    add('zeros', np.zeros((64, 256), np.float32))                   # one long run per stream (offset-1 matches)
    add('blocks_leftover', rng.integers(0, 2, (1000, 33)).astype(np.float32), blocksize=4096)  # 3 blocks, short last one
    add('blocks_exact', rng.integers(0, 2, (1024, 32)).astype(np.float32), blocksize=4096)     # 2 whole blocks
    add('noshuffle', relu[:8], shuffle=0)
    add('lz4hc', relu[:8], cname=b'lz4hc', clevel=9)
    add('zlib', relu[:8], cname=b'zlib')
    add('clevel1', rng.integers(0, 3, (32, 100)).astype(np.float32), clevel=1)
    add('clevel9', rng.integers(0, 3, (32, 100)).astype(np.float32), clevel=9)
    add('f8', rng.integers(0, 9, 3000).astype(np.float64))
    add('i2_odd', np.arange(10001, dtype=np.int16))                 # element count not a multiple of anything
    add('typesize24_nosplit', rng.integers(0, 3, (50, 240)).astype(np.uint8), typesize=24)   # typesize > 16: one stream
    add('typesize3_tail', rng.integers(0, 4, 1000).astype(np.uint8), typesize=3)             # 1000 % 3 != 0: unshuffled tail
    add('short_periods', np.tile(np.arange(3, dtype=np.uint8), 4000), typesize=1)            # matches with offset 3
    add('period7', np.tile(np.arange(7, dtype=np.uint8), 3000), shuffle=0, typesize=1)
    np.savez_compressed(os.path.join(OUT, 'frames.npz'), **frames)

    make_dataset_stores_and_loader_golden(rng)

    total = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(OUT) for f in fs)
    print(f'wrote {OUT}: {total / 1024:.0f} KiB, {len(expected)} arrays, {len(frames) // 2} frames')


if __name__ == '__main__':
    main()
