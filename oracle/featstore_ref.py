"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of how the reference reads its on-disk features: `zarr.open(path, mode='r')[video_id][name][:]`
(vhoi/data_loading.py:28,39-42,71-87,123-141) over the directory stores written by
`group.array(name, data, chunks=False, dtype=np.float32)` (vhoi/roi_features.py:227-242,292-295).

The algorithm lives in third-party dependencies that are absent from /root/reference and from this image:
zarr==2.4.0 and numcodecs==0.6.4 (reference environment.yml:113,125), the latter bundling c-blosc 1.x. Restated from
their published formats:
  * zarr storage spec v2: `.zgroup` / `.zarray` / `.zattrs` JSON documents, one file per chunk named by the chunk's
    grid index joined with '.', chunk bytes = compressor.encode(C- or F-ordered chunk buffer), edge chunks full-size,
    missing chunk = fill_value;
  * Blosc-1 chunk format (c-blosc README_CHUNK_FORMAT.rst + blosc.c blosc_d): header, block start table, split streams;
  * LZ4 block format (lz4_Block_format.md).
Pinned against frames produced by the REAL c-blosc 1.21.0 shared library that ships in this image's conda tree, driven
with numcodecs' exact call (blosc_compress_ctx(clevel=5, shuffle=1, typesize=itemsize, cname='lz4', blocksize=0)) --
golden G9, tools/make_golden_featstore.py. The zarr *directory layout* itself is pinned only to the spec (zarr is not
installable here): parity for that layer is "unpinned", stated in DESIGN.md.

Pure Python loops: use on small arrays only.
"""
import json
import os
import zlib

import numpy as np


def lz4_block_decode(src: bytes, expected: int) -> bytes:
    """LZ4 block -> bytes. Sequences: token, literals, 2-byte LE offset, match (length nibble + 4, may overlap)."""
    out = bytearray()
    i, n = 0, len(src)
    while True:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                s = src[i]; i += 1
                lit += s
                if s != 255:
                    break
        out += src[i:i + lit]
        if i + lit > n:
            raise ValueError('literals run past the block')
        i += lit
        if i == n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        if off == 0 or off > len(out):
            raise ValueError('bad match offset')
        ml = token & 15
        if ml == 15:
            while True:
                s = src[i]; i += 1
                ml += s
                if s != 255:
                    break
        ml += 4
        start = len(out) - off
        for k in range(ml):          # byte by byte: the match may run into bytes it is producing
            out.append(out[start + k])
    if len(out) != expected:
        raise ValueError(f'decoded {len(out)} bytes, expected {expected}')
    return bytes(out)


def blosc_decode(frame: bytes) -> bytes:
    """One Blosc-1 frame -> the bytes numcodecs.Blosc().decode would return."""
    version, _versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes = int.from_bytes(frame[4:8], 'little')
    blocksize = int.from_bytes(frame[8:12], 'little')
    cbytes = int.from_bytes(frame[12:16], 'little')
    if version != 2 or cbytes > len(frame):
        raise ValueError('not a Blosc-1 frame')
    if nbytes == 0:
        return b''
    if flags & 0x2:                                   # memcpyed
        return bytes(frame[16:16 + nbytes])
    if flags & 0x4:
        raise NotImplementedError('bit-shuffle')
    codec = (flags >> 5) & 7
    if codec not in (1, 3):
        raise NotImplementedError(f'inner codec {codec}')
    nblocks = -(-nbytes // blocksize)
    leftover = nbytes % blocksize
    out = bytearray()
    for j in range(nblocks):
        last_short = (j == nblocks - 1) and leftover > 0
        bsize = leftover if last_short else blocksize
        pos = int.from_bytes(frame[16 + 4 * j:20 + 4 * j], 'little')
        split = (not flags & 0x10) and typesize <= 16 and bsize // typesize >= 128 and not last_short
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        block = bytearray()
        for _ in range(nsplits):
            clen = int.from_bytes(frame[pos:pos + 4], 'little', signed=True); pos += 4
            payload = frame[pos:pos + clen]; pos += clen
            if clen == neblock:
                block += payload
            elif codec == 1:
                block += lz4_block_decode(payload, neblock)
            else:
                block += zlib.decompress(payload)
        if len(block) != bsize:
            raise ValueError('block size mismatch')
        if (flags & 0x1) and typesize > 1:            # undo the byte shuffle: plane j holds byte j of every element
            ne = bsize // typesize
            planes = np.frombuffer(bytes(block[:ne * typesize]), np.uint8).reshape(typesize, ne)
            block = bytearray(planes.T.tobytes()) + block[ne * typesize:]
        out += block
    return bytes(out)


def _decode_chunk(raw: bytes, compressor):
    if compressor is None:
        return raw
    cid = compressor['id']
    if cid == 'blosc':
        return blosc_decode(raw)
    if cid == 'zlib':
        return zlib.decompress(raw)
    raise NotImplementedError(cid)


def read_array(path: str) -> np.ndarray:
    """A zarr v2 array directory -> ndarray (what `zarr.open(path)[:]` returns)."""
    with open(os.path.join(path, '.zarray')) as f:
        meta = json.load(f)
    assert meta['zarr_format'] == 2 and not meta.get('filters')
    shape, chunks = tuple(meta['shape']), tuple(meta['chunks'])
    dtype = np.dtype(meta['dtype'])
    sep = meta.get('dimension_separator', '.')
    fill = meta['fill_value']
    out = np.empty(shape, dtype)
    grid = [range(-(-s // c)) for s, c in zip(shape, chunks)] if shape else []
    for idx in np.ndindex(*[len(g) for g in grid]) if shape else [()]:
        key = sep.join(str(i) for i in idx) if shape else '0'
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        fp = os.path.join(path, key)
        if not os.path.exists(fp):
            out[sel] = 0 if fill is None else fill
            continue
        with open(fp, 'rb') as f:
            raw = _decode_chunk(f.read(), meta['compressor'])
        chunk = np.frombuffer(raw, dtype).reshape(chunks, order=meta['order'])
        out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out


def read_path(root: str, key: str) -> np.ndarray:
    """`zarr.open(root)[key][:]` with key like 'video/Human1'."""
    return read_array(os.path.join(root, *key.split('/')))


def list_group(path: str):
    """Names of the members of a group directory (sorted, as zarr's DirectoryStore listdir does)."""
    assert os.path.exists(os.path.join(path, '.zgroup'))
    return sorted(n for n in os.listdir(path) if not n.startswith('.z') and os.path.isdir(os.path.join(path, n)))
