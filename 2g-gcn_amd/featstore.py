"""Reader (and plain writer) of the reference's on-disk feature stores, without the zarr / numcodecs dependency.

Drop-in for the slice of the zarr API the reference uses:

    reference                                                   here
    ---------------------------------------------------------   ------------------------------------------
    import zarr                                                 from twog_gcn_amd import featstore as zarr
    root = zarr.open(path, mode='r')            (data_loading.py:28,71,123,205,239)      same
    root[video_id]['Human1'][:]                 (:134-141, :80-87)                        same
    root[video_id + '/skeleton'][:]             (:39-42, :217-220)                        same
    video_id in root, name in group             (roi_features.py:226,236-240)             same
    zarr.group(store=zarr.DirectoryStore(p), overwrite=False)   (roi_features.py:206-207) same
    root.create_group(video_id); g.array(name, data, chunks=False, dtype=np.float32)      same (uncompressed)

Stores are zarr v2 directory stores: a `.zgroup` / `.zarray` JSON document per node and one file per chunk; the
reference writes every array as ONE chunk compressed with zarr's default Blosc(lz4, clevel 5, byte shuffle). Chunk
decoding is native code (include/twog_featstore.h -> libtwog_featstore.so, 2g-gcn_amd/csrc_host/featstore.c); this
module is the metadata / indexing layer. There is no Python decode fallback: a missing library raises.

MI355X-node specifics: `Array.read_into(buf)` decodes straight into a caller-owned buffer -- in practice a pinned
host tensor of the DevicePrefetcher (data_loading.py) -- so a clip's features go disk -> pinned staging -> HBM with
one host write and one DMA; `load_pinned` does that for a set of arrays.
"""
import builtins
import ctypes as C
import json
import os
import zlib

import numpy as np

from .hostcpu import effective_cpu_count

_open = builtins.open  # this module defines its own open(), like zarr does
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtwog_featstore.so')

ERRORS = {-1: 'bad argument', -2: 'not a Blosc-1 frame', -3: 'destination too small', -4: 'corrupt frame',
          -5: 'inner codec / filter not implemented (only lz4, lz4hc, zlib with byte shuffle or none)',
          -6: 'I/O error', -7: 'chunk decodes to a different size than the array metadata says', -8: 'out of memory'}

# C ABI (include/twog_featstore.h): name -> (restype, argtypes)
SIGNATURES = {
    'twog_fs_version': (C.c_char_p, []),
    'twog_lz4_block_decode': (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    'twog_blosc_info': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'twog_blosc_decode': (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    'twog_fs_read_chunk': (C.c_int64, [C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_int]),
}


class BloscInfo(C.Structure):  # twog_blosc_info_t
    _fields_ = [('version', C.c_int32), ('versionlz', C.c_int32), ('flags', C.c_int32), ('typesize', C.c_int32),
                ('nbytes', C.c_int64), ('blocksize', C.c_int64), ('cbytes', C.c_int64)]


_lib = None


def lib():
    """The native decoder; raises if it has not been built (make -C 2g-gcn_amd/csrc_host)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; '
                               f'g.build()"` (there is no Python fallback for chunk decoding)')
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


class FeatStoreError(RuntimeError):
    pass


def _check(rc, what):
    if rc < 0:
        raise FeatStoreError(f'{what}: {ERRORS.get(int(rc), rc)} (code {int(rc)})')
    return rc


def default_threads():
    return max(1, min(8, effective_cpu_count()))


def blosc_decode(frame, out=None, n_threads=None):
    """One Blosc-1 frame (bytes-like) -> uint8 ndarray (or into `out`, a writable C-contiguous uint8 buffer)."""
    src = np.frombuffer(frame, np.uint8)
    info = BloscInfo()
    _check(lib().twog_blosc_info(src.ctypes.data, src.nbytes, C.byref(info)), 'blosc header')
    if out is None:
        out = np.empty(info.nbytes, np.uint8)
    n = _check(lib().twog_blosc_decode(src.ctypes.data, src.nbytes, out.ctypes.data, out.nbytes,
                                       n_threads or default_threads()), 'blosc decode')
    return out[:n]


def blosc_info(frame):
    src = np.frombuffer(frame, np.uint8)
    info = BloscInfo()
    _check(lib().twog_blosc_info(src.ctypes.data, src.nbytes, C.byref(info)), 'blosc header')
    return {k: getattr(info, k) for k, _ in BloscInfo._fields_}


def lz4_block_decode(block, decoded_size):
    src = np.frombuffer(block, np.uint8)
    out = np.empty(decoded_size, np.uint8)
    n = _check(lib().twog_lz4_block_decode(src.ctypes.data, src.nbytes, out.ctypes.data, out.nbytes), 'lz4 block')
    return out[:n]


class DirectoryStore:
    """zarr.DirectoryStore stand-in: a path holder."""

    def __init__(self, path):
        self.path = os.fspath(path)


def _read_json(path):
    with _open(path) as f:
        return json.load(f)


def _as_host_array(out):
    """numpy view of a caller-owned destination (numpy array or CPU torch tensor, pinned or not)."""
    if isinstance(out, np.ndarray):
        return out
    if hasattr(out, 'numpy') and hasattr(out, 'is_contiguous'):
        if out.device.type != 'cpu':
            raise ValueError('read_into needs a host (CPU / pinned) tensor; copy to the device afterwards')
        return out.numpy()
    raise TypeError(f'unsupported destination {type(out)}')


class Array:
    """zarr.core.Array stand-in (read side): shape / dtype / chunks metadata and whole-array or indexed reads."""

    def __init__(self, path, name=''):
        self.path, self.name = path, name
        meta = _read_json(os.path.join(path, '.zarray'))
        if meta.get('zarr_format') != 2:
            raise FeatStoreError(f'{path}: zarr_format {meta.get("zarr_format")} (only v2 stores)')
        if meta.get('filters'):
            raise NotImplementedError(f'{path}: filters {meta["filters"]}')
        self.shape = tuple(meta['shape'])
        self.chunks = tuple(meta['chunks'])
        self.dtype = np.dtype(meta['dtype'])
        if self.dtype.hasobject or self.dtype.fields is not None:
            raise NotImplementedError(f'{path}: dtype {meta["dtype"]}')
        self.order = meta.get('order', 'C')
        self.fill_value = meta.get('fill_value')
        self.compressor = meta.get('compressor')
        self._sep = meta.get('dimension_separator', '.')
        cid = None if self.compressor is None else self.compressor.get('id')
        if cid not in (None, 'blosc', 'zlib', 'gzip'):
            raise NotImplementedError(f'{path}: compressor {cid!r} (implemented: none, blosc, zlib, gzip)')
        self._cid = cid

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))
    nbytes = property(lambda self: self.size * self.dtype.itemsize)
    attrs = property(lambda self: _attrs(self.path))

    def __len__(self):
        if not self.shape:
            raise TypeError('len() of a 0-d array')
        return self.shape[0]

    def __repr__(self):
        return f'<featstore.Array {self.name or self.path} {self.shape} {self.dtype}>'

    # -- chunk level ---------------------------------------------------------------------------------------------
    def _chunk_path(self, idx):
        return os.path.join(self.path, self._sep.join(map(str, idx)) if idx else '0')

    def _fill(self):
        return 0 if self.fill_value is None else self.fill_value

    def _decode_chunk_into(self, idx, dst, n_threads):
        """dst: C-contiguous ndarray of exactly one chunk's bytes (any dtype). Returns False if the chunk is absent."""
        fp = self._chunk_path(idx)
        if not os.path.exists(fp):
            return False
        if self._cid in (None, 'blosc'):
            _check(lib().twog_fs_read_chunk(fp.encode(), 0 if self._cid is None else 1, dst.ctypes.data, dst.nbytes,
                                            n_threads), fp)
        else:  # zarr-level zlib / gzip streams (not written by the reference): stdlib inflate, then one copy
            with _open(fp, 'rb') as f:
                raw = zlib.decompress(f.read(), 15 + 32)
            if len(raw) != dst.nbytes:
                raise FeatStoreError(f'{fp}: chunk decodes to {len(raw)} bytes, metadata says {dst.nbytes}')
            dst.reshape(-1).view(np.uint8)[:] = np.frombuffer(raw, np.uint8)
        return True

    def read_into(self, out, n_threads=None):
        """Decode the whole array into `out` (numpy array or host torch tensor of this shape and dtype, C-contiguous).
        Single-chunk C-order arrays -- everything the reference writes -- are decoded in place, no temporary."""
        dst = _as_host_array(out)
        if tuple(dst.shape) != self.shape or dst.dtype.itemsize != self.dtype.itemsize or not dst.flags.c_contiguous:
            raise ValueError(f'destination {dst.shape} {dst.dtype} does not match {self.shape} {self.dtype} (C order)')
        if dst.dtype != self.dtype and dst.dtype.newbyteorder() != self.dtype:
            raise ValueError(f'destination dtype {dst.dtype} is not {self.dtype}')
        nt = n_threads or default_threads()
        if self.size == 0:
            return out
        grid = tuple(-(-s // c) for s, c in zip(self.shape, self.chunks))
        if all(g == 1 for g in grid) and self.chunks == self.shape and self.order == 'C':
            if not self._decode_chunk_into((0,) * self.ndim, dst, nt):
                dst.view(self.dtype)[...] = self._fill()
        else:
            store_view = dst.view(self.dtype) if dst.dtype != self.dtype else dst
            tmp = np.empty(int(np.prod(self.chunks, dtype=np.int64)), self.dtype)
            for idx in np.ndindex(*grid):
                sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, self.chunks, self.shape))
                if not self._decode_chunk_into(idx, tmp, nt):
                    store_view[sel] = self._fill()
                    continue
                chunk = tmp.reshape(self.chunks, order=self.order)
                store_view[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
        if dst.dtype != self.dtype:  # destination has the other byte order: swap in place
            dst.view(self.dtype).byteswap(inplace=True)
        return out

    def _read_full(self):
        out = np.empty(self.shape, self.dtype)
        self.read_into(out)
        return out

    def __getitem__(self, sel):
        full = self._read_full()
        if sel is Ellipsis or (isinstance(sel, slice) and sel == slice(None)):
            return full
        return full[sel]

    def __array__(self, dtype=None, copy=None):
        a = self._read_full()
        return a if dtype is None else a.astype(dtype)


def _attrs(path):
    fp = os.path.join(path, '.zattrs')
    return _read_json(fp) if os.path.exists(fp) else {}


class Group:
    """zarr.hierarchy.Group stand-in over a directory."""

    def __init__(self, path, name='', read_only=True):
        self.path, self.name, self.read_only = path, name, read_only
        if not os.path.exists(os.path.join(path, '.zgroup')):
            raise FeatStoreError(f'{path}: not a zarr group (no .zgroup)')

    attrs = property(lambda self: _attrs(self.path))

    def _child(self, key):
        key = key.strip('/')
        p = os.path.join(self.path, *key.split('/')) if key else self.path
        return p, (f'{self.name}/{key}' if self.name else key)

    def __contains__(self, key):
        p, _ = self._child(key)
        return os.path.exists(os.path.join(p, '.zarray')) or os.path.exists(os.path.join(p, '.zgroup'))

    def __getitem__(self, key):
        p, name = self._child(key)
        if os.path.exists(os.path.join(p, '.zarray')):
            return Array(p, name)
        if os.path.exists(os.path.join(p, '.zgroup')):
            return Group(p, name, self.read_only)
        raise KeyError(key)

    def keys(self):
        return iter(self)

    def __iter__(self):
        for n in sorted(os.listdir(self.path)):
            if not n.startswith('.z') and n in self:
                yield n

    def __len__(self):
        return sum(1 for _ in self)

    def group_keys(self):
        return (n for n in self if os.path.exists(os.path.join(self.path, n, '.zgroup')))

    def array_keys(self):
        return (n for n in self if os.path.exists(os.path.join(self.path, n, '.zarray')))

    def __repr__(self):
        return f'<featstore.Group {self.name or self.path}>'

    # -- write side (plain: uncompressed single- or multi-chunk arrays; any zarr v2 reader opens them) -------------
    def _writable(self):
        if self.read_only:
            raise PermissionError('store opened read-only')

    def create_group(self, name):
        self._writable()
        p, full = self._child(name)
        if name in self:
            raise ValueError(f'{full} already exists')
        _make_group_dirs(self.path, name)
        return Group(p, full, False)

    def require_group(self, name):
        return self[name] if name in self else self.create_group(name)

    def array(self, name, data, chunks=False, dtype=None, **kwargs):
        """`group.array(name, data, chunks=False, dtype=np.float32)` (roi_features.py:227-242): one chunk holding the
        whole array. Written WITHOUT a compressor (`"compressor": null`): the Faster-RCNN features barely compress
        (~0.86 with the reference's Blosc/lz4) and an uncompressed chunk is read straight into pinned memory."""
        self._writable()
        if kwargs.get('compressor') is not None:
            raise NotImplementedError('this writer stores chunks uncompressed')
        a = np.ascontiguousarray(np.asarray(data, dtype=dtype))
        p, full = self._child(name)
        if name in self:
            raise ValueError(f'{full} already exists')
        if '/' in name.strip('/'):
            _make_group_dirs(self.path, name.strip('/').rsplit('/', 1)[0])
        os.makedirs(p, exist_ok=True)
        ch = a.shape if chunks in (False, None, True) else tuple(chunks)
        if len(ch) != a.ndim or any(c < 1 for c in ch if a.size):
            raise ValueError(f'chunks {ch} for shape {a.shape}')
        grid = tuple(-(-s // c) for s, c in zip(a.shape, ch)) if a.size else ()
        for idx in (np.ndindex(*grid) if a.size else ()):
            sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, ch, a.shape))
            chunk = np.zeros(ch, a.dtype)
            chunk[tuple(slice(0, s.stop - s.start) for s in sel)] = a[sel]
            with _open(os.path.join(p, '.'.join(map(str, idx)) if idx else '0'), 'wb') as f:
                f.write(chunk.tobytes())
        with _open(os.path.join(p, '.zarray'), 'w') as f:  # metadata last: a node exists once it is complete
            json.dump({'zarr_format': 2, 'shape': list(a.shape), 'chunks': list(ch), 'dtype': a.dtype.str,
                       'compressor': None, 'fill_value': 0 if a.dtype.kind in 'iub' else 0.0, 'order': 'C',
                       'filters': None}, f, indent=4, sort_keys=True)
        return Array(p, full)

    create_dataset = array


def _make_group_dirs(root, rel):
    p = root
    for part in rel.strip('/').split('/'):
        p = os.path.join(p, part)
        os.makedirs(p, exist_ok=True)
        if os.path.exists(os.path.join(p, '.zarray')):
            raise ValueError(f'{p} is an array')
        zg = os.path.join(p, '.zgroup')
        if not os.path.exists(zg):
            with _open(zg, 'w') as f:
                json.dump({'zarr_format': 2}, f)


def open(store, mode='r'):  # noqa: A001 (mirrors zarr.open)
    """zarr.open: a Group or an Array, depending on what lives at `store`. mode 'r' (default here, as in every
    reference call), 'r+' / 'a' (open or create a group for writing), 'w' is not offered (no overwrite)."""
    path = store.path if isinstance(store, DirectoryStore) else os.fspath(store)
    if mode not in ('r', 'r+', 'a'):
        raise ValueError(f"mode {mode!r}: use 'r', 'r+' or 'a'")
    if os.path.exists(os.path.join(path, '.zarray')):
        return Array(path)
    if os.path.exists(os.path.join(path, '.zgroup')):
        return Group(path, read_only=(mode == 'r'))
    if mode == 'a':
        return group(store=DirectoryStore(path))
    raise FeatStoreError(f'{path}: no zarr v2 group or array here')


def group(store=None, overwrite=False):
    """zarr.group(store=DirectoryStore(path), overwrite=False) (roi_features.py:206-207): open-or-create, writable."""
    if store is None:
        raise ValueError('a DirectoryStore (or path) is required: in-memory stores are not offered')
    if overwrite:
        raise NotImplementedError('overwrite=True')
    path = store.path if isinstance(store, DirectoryStore) else os.fspath(store)
    os.makedirs(path, exist_ok=True)
    zg = os.path.join(path, '.zgroup')
    if not os.path.exists(zg):
        if os.path.exists(os.path.join(path, '.zarray')):
            raise ValueError(f'{path} is an array')
        with _open(zg, 'w') as f:
            json.dump({'zarr_format': 2}, f)
    return Group(path, read_only=False)


def load_pinned(node, names, pin_memory=True, n_threads=None):
    """{name: host torch tensor} for the arrays `names` under group `node`, each decoded directly into (pinned) memory;
    native-endian copies of the stored dtype. Feed them to DevicePrefetcher / `.to(device, non_blocking=True)`."""
    import torch
    out = {}
    for n in names:
        arr = node[n]
        dt = arr.dtype.newbyteorder('=')
        t = torch.empty(arr.shape, dtype=getattr(torch, np.dtype(dt).name),
                        pin_memory=bool(pin_memory and torch.cuda.is_available()))
        arr.read_into(t, n_threads)
        out[n] = t
    return out

