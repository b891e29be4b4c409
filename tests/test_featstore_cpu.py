"""CPU: the feature-store reader against golden G9 -- stores and frames produced by the real c-blosc
1.21.0 with numcodecs' call (tools/make_golden_featstore.py) -- and against the oracle restatement.

  * oracle (oracle/featstore_ref.py) == golden                      -> the restatement of the formats is pinned
  * native decoder (libtwog_featstore.so through the C ABI) == golden, bit-exact, 1 and N threads
  * zarr-API mirror (twog_gcn_amd.featstore) reads the stores exactly as the reference's calls would
  * writer round trip; error behaviour; sanitizer build of the decoder under frame mutation
"""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from tests.helpers import ROOT

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import featstore as zarr   # the drop-in spelling
from twog_gcn_amd import featstore
from oracle import featstore_ref

G9 = os.path.join(ROOT, 'tests', 'golden', 'g9_featstore')
STORES = os.path.join(G9, 'stores')
HEADER = os.path.join(ROOT, 'include', 'twog_featstore.h')
HOST_SRC = os.path.join(ROOT, '2g-gcn_amd', 'csrc_host')


@pytest.fixture(scope='module')
def expected():
    return dict(np.load(os.path.join(G9, 'expected.npz')))


@pytest.fixture(scope='module')
def frames():
    z = np.load(os.path.join(G9, 'frames.npz'))
    return {k[6:]: (z[k], z['data_' + k[6:]]) for k in z.files if k.startswith('frame_')}


def _same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


# ---------------------------------------------------------------------------------------------- oracle vs golden
def test_oracle_decodes_every_golden_frame(frames):
    assert len(frames) >= 17
    for name, (frame, data) in frames.items():
        assert featstore_ref.blosc_decode(frame.tobytes()) == data.tobytes(), name


def test_oracle_reads_every_golden_array(expected):
    assert len(expected) >= 23
    for key, want in expected.items():
        store, path = key.split('|')
        assert _same(featstore_ref.read_path(os.path.join(STORES, store), path), want), key


# ---------------------------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r'\b(twog_[a-z0-9_]+)\s*\(', src)))
    assert set(names) == set(featstore.SIGNATURES), set(names) ^ set(featstore.SIGNATURES)
    h = ctypes.CDLL(featstore.LIB_PATH)
    for n in names:
        assert hasattr(h, n), n
    assert b'blosc' in featstore.lib().twog_fs_version()


def test_info_struct_layout_matches_the_c_compiler(tmp_path):
    src = tmp_path / 'sz.c'
    src.write_text(f'#include <stdio.h>\n#include "{HEADER}"\nint main(void){{printf("%zu\\n", sizeof(twog_blosc_info_t));return 0;}}\n')
    subprocess.run(['gcc', str(src), '-o', str(tmp_path / 'sz')], check=True)
    out = subprocess.run([str(tmp_path / 'sz')], capture_output=True, text=True, check=True).stdout
    assert int(out) == ctypes.sizeof(featstore.BloscInfo)


def test_reader_fails_loudly_without_the_library(monkeypatch):
    monkeypatch.setattr(featstore, 'LIB_PATH', os.path.join(ROOT, 'does_not_exist.so'))
    monkeypatch.setattr(featstore, '_lib', None)
    with pytest.raises(RuntimeError, match='no Python fallback'):
        zarr.open(os.path.join(STORES, 'features.zarr'))['Subject14-Cheering-1/Human1'][:]


# ---------------------------------------------------------------------------------------------- native decoder
@pytest.mark.parametrize('threads', [1, 4])
def test_native_decoder_matches_golden_frames(frames, threads):
    for name, (frame, data) in frames.items():
        got = featstore.blosc_decode(frame, n_threads=threads)
        assert got.tobytes() == data.tobytes(), name
        info = featstore.blosc_info(frame)
        assert info['nbytes'] == data.nbytes and info['cbytes'] == frame.nbytes and info['version'] == 2


def test_frame_flags_cover_the_container_format(frames):
    """The golden set exercises: split + shuffle, memcpyed, no shuffle, do-not-split (typesize > 16), zlib inner codec,
    several blocks with and without a short last block, an unshuffled element tail."""
    f = {n: featstore.blosc_info(fr) for n, (fr, _) in frames.items()}
    assert f['default_f4']['flags'] == 0x21
    assert f['tiny_memcpyed']['flags'] & 0x2 and f['incompressible_u1']['flags'] & 0x2
    assert f['noshuffle']['flags'] & 0x1 == 0
    assert f['typesize24_nosplit']['flags'] & 0x10
    assert f['zlib']['flags'] >> 5 == 3
    assert f['blocks_leftover']['nbytes'] > 2 * f['blocks_leftover']['blocksize']
    assert f['blocks_leftover']['nbytes'] % f['blocks_leftover']['blocksize'] != 0
    assert f['blocks_exact']['nbytes'] == 2 * f['blocks_exact']['blocksize']
    assert f['typesize3_tail']['nbytes'] % 3 != 0


def test_lz4_block_known_answers():
    # hand-assembled blocks (LZ4 block format): literals only; literal + overlapping match (run); long literal length
    assert featstore.lz4_block_decode(bytes([0x50]) + b'hello', 5).tobytes() == b'hello'
    blk = bytes([0x1f, ord('a'), 0x01, 0x00, 0x05, 0x10, ord('b')])   # 'a', match off=1 len 15+5+4=24, then literal 'b'
    assert featstore.lz4_block_decode(blk, 26).tobytes() == b'a' * 25 + b'b'
    assert featstore_ref.lz4_block_decode(blk, 26) == b'a' * 25 + b'b'
    lit = bytes(range(256)) + bytes(range(44))
    blk = bytes([0xf0, 255, 30]) + lit                                  # 15 + 255 + 30 = 300 literals
    assert featstore.lz4_block_decode(blk, 300).tobytes() == lit
    blk = bytes([0x32, 1, 2, 3, 0x03, 0x00, 0x00])                      # '123' then match off=3 len 6, then empty literals
    assert featstore.lz4_block_decode(blk, 9).tobytes() == bytes([1, 2, 3] * 3)


def test_decoder_rejects_malformed_input(frames):
    frame, data = frames['default_f4']
    with pytest.raises(featstore.FeatStoreError, match='not a Blosc-1 frame'):
        featstore.blosc_decode(frame[:10])
    bad = frame.copy(); bad[0] = 9
    with pytest.raises(featstore.FeatStoreError, match='not a Blosc-1 frame'):
        featstore.blosc_decode(bad)
    with pytest.raises(featstore.FeatStoreError):
        featstore.blosc_decode(frame[:frame.nbytes // 2])                # cbytes > available
    with pytest.raises(featstore.FeatStoreError, match='too small'):
        featstore.blosc_decode(frame, out=np.empty(data.nbytes - 1, np.uint8))
    bad = frame.copy(); bad[2] |= 0x4                                    # bit-shuffle flag
    with pytest.raises(featstore.FeatStoreError, match='not implemented'):
        featstore.blosc_decode(bad)
    bad = frame.copy(); bad[2] = (bad[2] & 0x1f) | (4 << 5)              # inner codec zstd
    with pytest.raises(featstore.FeatStoreError, match='not implemented'):
        featstore.blosc_decode(bad)
    bad = frame.copy(); bad[16:20] = np.frombuffer(np.uint32(5).tobytes(), np.uint8)   # block start inside the header
    with pytest.raises(featstore.FeatStoreError, match='corrupt'):
        featstore.blosc_decode(bad)
    with pytest.raises(featstore.FeatStoreError, match='corrupt|too small'):
        featstore.lz4_block_decode(bytes([0x1f, ord('a'), 0x02, 0x00, 0x00]), 64)     # offset 2 with 1 byte produced


def test_sanitizer_build_survives_frame_mutation(frames, tmp_path):
    """ASan/UBSan build of the decoder (CPU only), a few thousand corrupted copies of representative frames."""
    subprocess.run(['make', '-C', HOST_SRC, 'featstore_fuzz'], check=True, capture_output=True)
    for i, name in enumerate(('default_f4', 'zeros', 'blocks_leftover', 'typesize3_tail', 'zlib', 'tiny_memcpyed')):
        p = tmp_path / name
        frames[name][0].tofile(p)
        r = subprocess.run([os.path.join(HOST_SRC, 'featstore_fuzz'), str(p), '1500', str(i)], capture_output=True,
                           text=True)
        assert r.returncode == 0 and r.stdout.startswith('ok '), (name, r.returncode, r.stderr[-2000:])
        decoded, rejected = map(int, r.stdout.split()[1:])
        assert decoded + rejected == 1500 and rejected > 0


# ---------------------------------------------------------------------------------------------- zarr API mirror
def test_reference_read_calls(expected):
    """The exact call shapes of vhoi/data_loading.py:28-42,123-141."""
    root = zarr.open(os.path.join(STORES, 'features.zarr'), mode='r')
    vids = [v for v in root]
    assert vids == ['Subject14-Cheering-1', 'Subject25-Co_working-3'] == featstore_ref.list_group(root.path)
    for vid in vids:
        assert vid in root and 'nope' not in root
        for name in ('Human1', 'Human2', 'objects', 'Human1_bbs', 'objects_bbs', 'Human1_pose'):
            want = expected[f'features.zarr|{vid}/{name}']
            assert _same(root[vid][name][:], want)                      # root[video_id]['Human1'][:]
            assert _same(root[vid + '/' + name][:], want)               # root[video_id + '/skeleton'][:]
            assert _same(np.asarray(root[vid][name]), want)
            assert name in root[vid]
    a = root['Subject14-Cheering-1']['objects']
    assert a.shape == (10, 3, 256) and a.dtype == np.float32 and a.chunks == a.shape and len(a) == 10
    assert a.compressor == {'id': 'blosc', 'cname': 'lz4', 'clevel': 5, 'shuffle': 1, 'blocksize': 0}
    want = expected['features.zarr|Subject14-Cheering-1/objects']
    assert _same(a[2:5, 1], want[2:5, 1]) and _same(a[...], want) and a[3, 2, 7] == want[3, 2, 7]
    assert root['Subject14-Cheering-1'].attrs == {'fps': 30, 'note': 'golden'}
    assert sorted(root['Subject14-Cheering-1'].array_keys()) == sorted(
        ['Human1', 'Human2', 'objects', 'Human1_bbs', 'objects_bbs', 'Human1_pose'])
    assert list(root.group_keys()) == vids
    with pytest.raises(KeyError):
        root['Subject14-Cheering-1']['Human3']
    with pytest.raises(PermissionError):
        root.create_group('x')


def test_format_variants_match_golden_and_oracle(expected):
    root = zarr.open(os.path.join(STORES, 'variants.zarr'))
    keys = [k.split('|')[1] for k in expected if k.startswith('variants.zarr|')]
    assert len(keys) >= 11
    for key in keys:
        want = expected[f'variants.zarr|{key}']
        got = root[key][:]
        assert _same(got, want), key
        assert _same(got, featstore_ref.read_path(root.path, key)), key
    # a destination in the other byte order gets native values
    be = root['big_endian']
    out = np.empty(be.shape, np.float32)
    be.read_into(out)
    assert np.array_equal(out, expected['variants.zarr|big_endian'].astype(np.float32))
    miss = np.empty((13, 50), '>f4')
    root['g/grid_raw_missing'].read_into(miss.view('<f4'))
    assert np.array_equal(miss.view('<f4'), expected['variants.zarr|g/grid_raw_missing'])


def test_read_into_host_tensor_and_checks(expected):
    import torch
    root = zarr.open(os.path.join(STORES, 'features.zarr'))
    a = root['Subject25-Co_working-3/objects']
    t = torch.full(a.shape, -1.0)
    assert a.read_into(t) is t
    assert np.array_equal(t.numpy(), expected['features.zarr|Subject25-Co_working-3/objects'])
    with pytest.raises(ValueError, match='does not match'):
        a.read_into(torch.empty(10, 3, 255))
    with pytest.raises(ValueError, match='does not match'):
        a.read_into(torch.empty(10, 256, 3).transpose(1, 2))            # right shape, not contiguous
    with pytest.raises(ValueError, match='dtype'):
        a.read_into(torch.empty(a.shape, dtype=torch.int32))
    got = featstore.load_pinned(root['Subject25-Co_working-3'], ['Human1', 'Human1_bbs'], pin_memory=False)
    assert np.array_equal(got['Human1'].numpy(), expected['features.zarr|Subject25-Co_working-3/Human1'])
    assert got['Human1_bbs'].dtype == torch.float32


def test_chunk_size_mismatch_is_an_error(tmp_path, expected):
    import shutil
    dst = tmp_path / 's.zarr'
    shutil.copytree(os.path.join(STORES, 'features.zarr'), dst)
    meta_p = dst / 'Subject14-Cheering-1' / 'Human1' / '.zarray'
    meta_p.write_text(meta_p.read_text().replace('256', '128'))         # metadata now claims half the bytes
    with pytest.raises(featstore.FeatStoreError, match='different size'):
        zarr.open(str(dst))['Subject14-Cheering-1/Human1'][:]
    raw = tmp_path / 'r.zarr'
    shutil.copytree(os.path.join(STORES, 'variants.zarr'), raw)
    with open(raw / 'raw' / '0.0', 'ab') as f:
        f.write(b'\0')
    with pytest.raises(featstore.FeatStoreError, match='different size'):
        zarr.open(str(raw))['raw'][:]
    with pytest.raises(featstore.FeatStoreError, match='no zarr v2'):
        zarr.open(str(tmp_path / 'missing.zarr'))


def test_writer_round_trip_in_the_reference_call_shape(tmp_path):
    """roi_features.py:206-207,226-242: group(store=DirectoryStore) / create_group / array(chunks=False, dtype=f32)."""
    rng = np.random.default_rng(3)
    store = zarr.DirectoryStore(str(tmp_path / 'out.zarr'))
    root = zarr.group(store=store, overwrite=False)
    data = {vid: {'skeleton': rng.standard_normal((7, 2048)), 'objects': rng.standard_normal((7, 5, 2048))}
            for vid in ('v1', 'v2')}
    for vid, arrays in data.items():
        if vid not in root:
            g = root.create_group(vid)
            for name, a in arrays.items():
                g.array(name, a, chunks=False, dtype=np.float32)
    again = zarr.group(store=store, overwrite=False)                     # reopen: nothing is recreated
    assert 'v1' in again and 'skeleton' in again['v1']
    with pytest.raises(ValueError, match='already exists'):
        again.create_group('v1')
    ro = zarr.open(store.path, mode='r')
    for vid, arrays in data.items():
        for name, a in arrays.items():
            got = ro[vid][name][:]
            assert got.dtype == np.float32 and np.array_equal(got, a.astype(np.float32))
            assert _same(got, featstore_ref.read_path(store.path, f'{vid}/{name}'))   # a spec reader agrees
    assert ro['v1']['skeleton'].compressor is None and ro['v1']['skeleton'].chunks == (7, 2048)
    # chunk-grid writes and nested names
    again.array('deep/er/x', np.arange(35, dtype=np.int64).reshape(5, 7), chunks=(2, 3))
    assert np.array_equal(ro['deep']['er']['x'][:], np.arange(35).reshape(5, 7))
    assert _same(ro['deep/er/x'][:], featstore_ref.read_path(store.path, 'deep/er/x'))
