"""Tensor-level interface to the gfx950 kernels (lib2ggcn_hip.so).

Every method takes torch CUDA fp32 tensors (or strided views of them), turns them into raw device pointers + strides
and calls ONE C-ABI entry point of include/twog_gcn.h on torch's current HIP stream. torch is used only to own device
memory and streams; no torch math runs here. There is no CPU / eager fallback: on a box without the built library
``HipKernels()`` raises.

Row-strided views: a matrix operand may be a 2-D view (rows, cols) or a 3-D view (outer, inner, cols) with unit
stride on cols -- it maps onto twog_rows_t without a copy.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_CUR_DEVICE = getattr(torch._C, '_cuda_getDevice', None)
if _CUR_DEVICE is None:
    _RAW_STREAM = None


def rows_of(t):
    """twog_rows_t of a 2-D (rows, cols) or 3-D (outer, inner, cols) fp32 view with unit column stride."""
    r = L.Rows()
    if t is None:
        r.ptr, r.ld_outer, r.ld_inner, r.inner = 0, 0, 0, 1
        return r
    assert t.dtype == torch.float32, t.dtype
    assert t.dim() in (2, 3), t.shape
    assert t.shape[-1] == 1 or t.stride(-1) == 1, (t.shape, t.stride())
    r.ptr = t.data_ptr()
    if t.dim() == 2:
        r.inner, r.ld_outer, r.ld_inner = 1, t.stride(0), t.stride(0)
    else:
        r.inner, r.ld_outer, r.ld_inner = t.shape[1], t.stride(0), t.stride(1)
        if t.shape[1] == 1:
            r.inner, r.ld_inner = 1, t.stride(0)
    return r


def fill_rows(r, t):
    """rows_of(t) written into the existing twog_rows_t `r` (a field of a descriptor array: no temporary, no copy)."""
    assert t.dtype == torch.float32, t.dtype
    nd = t.dim()
    st = t.stride()
    assert nd in (2, 3) and (t.shape[-1] == 1 or st[-1] == 1), (t.shape, st)
    r.ptr = t.data_ptr()
    if nd == 2 or t.shape[1] == 1:
        r.inner, r.ld_outer, r.ld_inner = 1, st[0], st[0]
    else:
        r.inner, r.ld_outer, r.ld_inner = t.shape[1], st[0], st[1]


def n_rows(t):
    return t.shape[0] if t.dim() == 2 else t.shape[0] * t.shape[1]


class HipKernels:
    """The product backend. `tests/fake_kernels.py` implements the same methods in plain torch for CPU-side tests of
    the host logic; it is never reachable from this package."""

    name = 'hip'

    def __init__(self):
        self.lib = L.load()
        self._ws = {}
        self._tape = None

    # ---------------------------------------------------------------- utilities
    def _stream(self):
        # the raw handle of torch's current stream on the current device: two C calls (~0.3 us) instead of the
        # torch.cuda.current_stream() Stream object (~8 us) -- a host-composed step asks for it thousands of times
        if self._tape is not None:
            # Between tape_begin() and tape_end() only the recordable entry points may be called (they return before they get
            # here): anything else would launch NOW, for the composed steps only, and be missing from every replayed step --
            # and the replay could not know. A programming error, raised where it happens.
            raise RuntimeError('a kernel entry point that cannot be recorded was called while a step is being taped '
                               '(kernels.tape_begin): only gemm (not chain), gru_step_*, relation_*_many and rowops may run there')
        if _RAW_STREAM is not None:
            return _RAW_STREAM(_CUR_DEVICE())
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError(f'{what} failed with code {rc}')

    def empty(self, *shape, like=None, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=like.device)

    def zeros(self, *shape, like=None, dtype=torch.float32, device=None):
        """A zeroed buffer: torch owns the memory, the library clears it (twog_fill_zero) -- no ATen kernel on the path."""
        t = torch.empty(*shape, dtype=dtype, device=like.device if like is not None else device)
        return self.fill_zero(t)

    def fill_zero(self, t):
        """In-place clear of a contiguous tensor (the whole storage range it spans)."""
        assert t.is_contiguous()
        if t.numel():
            self._check(self.lib.twog_fill_zero(t.data_ptr(), t.numel() * t.element_size(), self._stream()), 'twog_fill_zero')
        return t

    # ---------------------------------------------------------------- side stream (independent work beside a launch chain)
    class _Side:
        """`with side:` issues launches on a second stream of the device; it starts behind everything the caller's stream
        holds at construction, join() makes the caller's stream wait for it. Tensors allocated inside belong to the side
        stream's pool (torch's allocator handles their reuse); tensors of the caller's stream that the side launches read must
        stay referenced until join() (ops keeps them in the saved state of the pass)."""

        def __init__(self, stream):
            self.main = torch.cuda.current_stream()
            self.stream = stream
            ev = torch.cuda.Event()
            ev.record(self.main)
            stream.wait_event(ev)
            self._ctx = None

        def __enter__(self):
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
            return self

        def __exit__(self, *a):
            return self._ctx.__exit__(*a)

        def join(self):
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.main.wait_event(ev)

    _side_streams = {}

    def side_stream(self, dev, cus=0):
        """cus > 0: a stream restricted to that many compute units (twog_stream_create_masked: cus / 8 on each XCD), so that
        what runs on it leaves the other CUs to the caller's stream; None when the runtime refuses the mask. cus = 0: an
        ordinary second stream."""
        key = (self._dev_index(dev), int(cus))
        st = HipKernels._side_streams.get(key)
        if st is None:
            if cus > 0:
                with torch.cuda.device(dev):
                    h = C.c_void_p()
                    rc = self.lib.twog_stream_create_masked(int(cus), C.byref(h))
                st = torch.cuda.ExternalStream(h.value, device=dev) if rc == 0 and h.value else False
            elif os.environ.get('TWOG_SIDE_PRIORITY', 'normal') == 'low':
                # (experiment, round 6: the side stream's GEMM workgroups at the lowest queue priority, so that the chain's
                # workgroups on the caller's stream win free compute units; profiles/r06_side_stream_priority_ab.txt)
                with torch.cuda.device(dev):
                    h = C.c_void_p()
                    rc = self.lib.twog_stream_create_low_priority(C.byref(h))
                st = torch.cuda.ExternalStream(h.value, device=dev) if rc == 0 and h.value else torch.cuda.Stream(device=dev)
            else:
                st = torch.cuda.Stream(device=dev)
            HipKernels._side_streams[key] = st
        return HipKernels._Side(st) if st else None

    def debug_occupy(self, n_blocks, lds_bytes, usec):
        """Diagnostics (tests): n_blocks workgroups holding lds_bytes of LDS each for usec microseconds on the current stream."""
        self._check(self.lib.twog_debug_occupy(n_blocks, lds_bytes, usec, self._stream()), 'twog_debug_occupy')

    def copy_blocks(self, pairs):
        """pairs: (src, dst) contiguous fp32 tensors of equal element count; dst[...] = src[...] for all of them in ONE
        launch (twog_copy_blocks) -- the per-forward build of the packed operands (ops.pack_weights)."""
        pairs = [(s, d) for s, d in pairs if s.numel()]
        for i in range(0, len(pairs), L.COPY_MAX):
            chunk = pairs[i:i + L.COPY_MAX]
            arr = (L.Copy * len(chunk))()
            for a, (s, d) in zip(arr, chunk):
                assert s.dtype == torch.float32 and d.dtype == torch.float32 and s.is_contiguous() and d.is_contiguous()
                assert s.numel() == d.numel() and s.device == d.device
                a.src, a.dst, a.n = s.data_ptr(), d.data_ptr(), s.numel()
            self._check(self.lib.twog_copy_blocks(arr, len(chunk), self._stream()), 'twog_copy_blocks')

    def zeros_many(self, shapes, device, dtype=torch.float32):
        """Zeroed fp32 buffers of the given shapes carved out of ONE allocation and cleared by ONE launch (each view starts
        on a 256-byte boundary)."""
        sizes = [int(math.prod(sh)) for sh in shapes]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) // 64 * 64
        buf = self.zeros(max(total, 1), dtype=dtype, device=device)
        return [buf[o:o + n].view(*sh) for o, n, sh in zip(offs, sizes, shapes)]

    def workspace(self, nbytes, device, key='ws'):
        """A persistent scratch buffer per (device, key, current stream); grown on demand, contents undefined between
        calls. Per stream because launches on different streams may run concurrently (split-K slabs, reduction partials);
        launches that share a buffer are ordered by their stream."""
        k = (str(device), key, int(self._stream() or 0))
        buf = self._ws.get(k)
        if buf is None or buf.numel() * 4 < nbytes:
            buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
            if key == 'splitk':
                # the first 16 KB of the split-K workspace are the arrival tickets of the in-launch combine (include/twog_gcn.h,
                # twog_gemm_f32): zero when first handed over, returned to zero by every launch
                self.fill_zero(buf[:4096])
            self._ws_put(k, buf)
        else:
            self._ws_touch(k)
        return buf

    MAX_STREAMS_PER_WORKSPACE = 4

    def _ws_touch(self, k):
        self._ws[k] = self._ws.pop(k)   # most recently used last (dict order)

    def _ws_put(self, k, buf):
        """Workspaces are per (device, kind, stream): at most MAX_STREAMS_PER_WORKSPACE streams keep one of each kind, the
        least recently used entry goes first (a 320 MB split-K buffer + a 48 MB chain workspace per stream that ever
        issued a GEMM would otherwise stay for the life of the process). Dropping an entry only drops torch's reference:
        the caching allocator keeps the memory valid for work already enqueued on the stream that used it."""
        self._ws.pop(k, None)
        self._ws[k] = buf
        same = [q for q in self._ws if q[0] == k[0] and q[1] == k[1]]
        for q in same[:-self.MAX_STREAMS_PER_WORKSPACE]:
            del self._ws[q]

    def release_workspaces(self):
        """Drops every cached workspace (they are re-created on demand)."""
        self._ws.clear()

    def chain_workspace(self, device):
        """(pointer, bytes) of the chain workspace of the CURRENT stream on `device` (include/twog_gcn.h,
        twog_gemm_f32_chain): arrival tickets (zeroed here, once; every launch returns them to zero) + room for the
        partial tiles of a reduction split over workgroups. One buffer per (device, stream): launches that share it are
        ordered by the stream."""
        k = (str(device), 'chain', int(self._stream() or 0))
        buf = self._ws.get(k)
        if buf is None:
            buf = torch.zeros(int(self.lib.twog_chain_workspace_bytes()) // 4, dtype=torch.float32, device=device)
            self._ws_put(k, buf)
        return buf.data_ptr(), buf.numel() * 4

    def version(self):
        return self.lib.twog_version().decode()

    # ---------------------------------------------------------------- GEMM
    def _gemm_array(self, problems, a_kmajor, b_kmajor):
        n = len(problems)
        arr = (L.Gemm * n)()
        for i, p in enumerate(problems):
            A, B, Cm = p['A'], p['B'], p['C']
            g = arr[i]
            fill_rows(g.A, A)
            fill_rows(g.B, B)
            fill_rows(g.C, Cm)
            if a_kmajor:
                K, M = n_rows(A), A.shape[-1]
            else:
                M, K = n_rows(A), A.shape[-1]
            if b_kmajor:
                Kb, N = n_rows(B), B.shape[-1]
            else:
                N, Kb = n_rows(B), B.shape[-1]
            assert K == Kb, (A.shape, B.shape, a_kmajor, b_kmajor)
            assert n_rows(Cm) == M and Cm.shape[-1] == N, (Cm.shape, M, N)
            bias = p.get('bias')
            if bias is not None:
                assert bias.numel() == N and bias.is_contiguous()
            g.bias = _ptr(bias)
            g.M, g.N, g.K = M, N, K
            g.act = int(p.get('act', 0))
            g.accumulate = int(bool(p.get('accumulate', False)))
            b = p.get('batch')
            if b is None:
                g.batch, g.a_batch_stride, g.b_batch_stride, g.c_batch_stride = 1, 0, 0, 0
            else:
                g.batch, g.a_batch_stride, g.b_batch_stride, g.c_batch_stride = b
            cs = p.get('colsum')
            if cs is not None:
                assert a_kmajor and cs.numel() == M and cs.is_contiguous() and cs.dtype == torch.float32
            g.a_colsum = _ptr(cs)
            g.a_colsum_accumulate = int(bool(p.get('colsum_accumulate', False)))
        return arr

    def gemm_colsum_ok(self, problem):
        """True if gemm([problem], a_kmajor=True, b_kmajor=True) with a 'colsum' entry takes the column sums of A inside the
        GEMM launch (twog_gemm_colsum_fused); False: the caller takes them with colsum()."""
        key = tuple((tuple(t.shape), tuple(t.stride()), t.data_ptr() % 16) for t in (problem['A'], problem['B'], problem['C']))
        ok = self._colsum_ok.get(key)
        if ok is None:
            arr = self._gemm_array([dict(problem, colsum=None)], True, True)
            arr[0].a_colsum = arr[0].C.ptr   # (any non-null pointer: the query launches nothing)
            ws = self.workspace(320 << 20, problem['C'].device, 'splitk')
            ok = self._colsum_ok[key] = bool(self.lib.twog_gemm_colsum_fused(arr, 1, 1, 1, ws.data_ptr(), ws.numel() * 4))
        return ok

    _colsum_ok = {}

    def gemm(self, problems, a_kmajor=False, b_kmajor=False, split_k_workspace=True, chain=False):
        """problems: list of dicts with keys A, B, C (row-strided views), bias (1-D or None), act (0/1), accumulate.
        Logical shapes: A (M,K) [or stored (K,M) if a_kmajor], B (N,K) [or (K,N) if b_kmajor], C (M,N).
        Optional 'batch': (n, a_stride, b_stride, c_stride) in elements. chain=True: a launch of a recurrent chain
        (twog_gemm_f32_chain: the reduction may be split over workgroups and combined inside the launch).
        Optional 'colsum' ([M] fp32, with 'colsum_accumulate'): the column sums of a k-major A from the same pass over A
        (twog_gemm_t::a_colsum; only where gemm_colsum_ok(problem) holds)."""
        n = len(problems)
        if n == 0:
            return
        dev = problems[0]['C'].device
        arr = self._gemm_array(problems, a_kmajor, b_kmajor)
        if chain:
            assert self._tape is None, 'chain launches cannot be recorded (kernels.tape_begin)'
            rc = self.lib.twog_gemm_f32_chain(arr, n, int(a_kmajor), int(b_kmajor), *self.chain_workspace(dev), self._stream())
            self._check(rc, 'twog_gemm_f32_chain')
            return
        if self._tape is not None:
            self._tape.append((L.TAPE_GEMM, n, int(a_kmajor) | int(b_kmajor) << 1, arr))
            return
        ws_ptr, ws_bytes = 0, 0
        if split_k_workspace:
            ws = self.workspace(320 << 20, dev, 'splitk')
            ws_ptr, ws_bytes = ws.data_ptr(), ws.numel() * 4
        rc = self.lib.twog_gemm_f32(arr, n, int(a_kmajor), int(b_kmajor), ws_ptr, ws_bytes, self._stream())
        self._check(rc, 'twog_gemm_f32')

    GEMM_TILE128, GEMM_WAVES8, GEMM_KG, GEMM_SPLITK, GEMM_GATE, GEMM_KSPLIT, GEMM_GRUFWD, GEMM_ROWS32 = 1, 2, 4, 8, 16, 32, 64, 128  # TWOG_GEMM_CLASS_*
    GEMM_XSPLIT = 256
    GEMM_X3 = 512

    def gemm_last_class(self):
        """Bit field (GEMM_*) of the kernel variant the most recent gemm() chunk of this thread selected."""
        return int(self.lib.twog_gemm_last_class())

    # ---------------------------------------------------------------- geometric-level GCN
    @staticmethod
    def _geo(x_human):
        """x_human (bs, T, H, 2048+4N) contiguous -> (pointer to geometry of human 0 of frame 0, frame stride, frames)."""
        assert x_human.is_contiguous() and x_human.dtype == torch.float32
        bs, T, H, Fh = x_human.shape
        return x_human.data_ptr() + 2048 * 4, H * Fh, bs * T

    def bn_fold(self, x_human, n_nodes, gamma, beta, running_mean, running_var, num_batches_tracked, training,
                stats_reduce=None, fold=None):
        """stats_reduce (optional, sync-BN of distributed.DataParallel): callable (sums fp64 [2*4N], n_frames) ->
        (sums reduced over the ranks, total frames); applied to the batch statistics before they are folded.
        fold = (wq [128,64], wk [128,64], bq [128]): the same launch also folds the similarity projections into
        md [65,64] = [Mt | d] (see twog_bn_finalize); then returns (ab, mi, md)."""
        ptr, fstride, nf = self._geo(x_human)
        nch = 4 * n_nodes
        dev = x_human.device
        ab = torch.empty(2, nch, dtype=torch.float32, device=dev)
        mi = torch.empty(2, nch, dtype=torch.float32, device=dev)
        nblk = max(1, min(240, nf // 8))   # >= 8 frames per block, at most 240 blocks (the finalize kernel sums them 4-wide)
        partials = torch.empty(nblk * 2 * nch, dtype=torch.float64, device=dev)
        if training:
            self._check(self.lib.twog_bn_stats(ptr, fstride, nf, n_nodes, partials.data_ptr(), nblk, self._stream()),
                        'twog_bn_stats')
            if stats_reduce is not None:
                partials, nf = stats_reduce(partials.view(nblk, 2 * nch).sum(0), nf)
                partials, nblk = partials.contiguous(), 1
        md = torch.empty(65, 64, dtype=torch.float32, device=dev) if fold is not None else None
        wq, wk, bq = fold if fold is not None else (None, None, None)
        self._check(self.lib.twog_bn_finalize(partials.data_ptr(), nblk, nf, n_nodes, gamma.data_ptr(),
                                              beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                              _ptr(num_batches_tracked), int(training), ab.data_ptr(), mi.data_ptr(),
                                              _ptr(wq), _ptr(wk), _ptr(bq), _ptr(md), self._stream()), 'twog_bn_finalize')
        return (ab, mi) if fold is None else (ab, mi, md)

    def gcn_embed1_fwd(self, x_human, n_nodes, ab, w1, b1):
        ptr, fstride, nf = self._geo(x_human)
        e1 = torch.empty(nf * n_nodes, 64, dtype=torch.float32, device=x_human.device)
        self._check(self.lib.twog_gcn_embed1_fwd(ptr, fstride, nf, n_nodes, ab.data_ptr(), w1.data_ptr(),
                                                 b1.data_ptr(), e1.data_ptr(), self._stream()), 'twog_gcn_embed1_fwd')
        return e1

    def gcn_fused_fwd(self, x_human, n_nodes, ab, w1, b1, w2, b2, md, save_x=True):
        """Geo_gcn forward up to the aggregation in one kernel: returns (X or None, adj [F,N,N], Z [(f,n),64])."""
        ptr, fstride, nf = self._geo(x_human)
        dev = x_human.device
        X = torch.empty(nf * n_nodes, 64, dtype=torch.float32, device=dev) if save_x else None
        adj = torch.empty(nf, n_nodes, n_nodes, dtype=torch.float32, device=dev)
        Z = torch.empty(nf * n_nodes, 64, dtype=torch.float32, device=dev)
        self._check(self.lib.twog_gcn_fused_fwd(ptr, fstride, nf, n_nodes, ab.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                                w2.data_ptr(), b2.data_ptr(), md.data_ptr(), _ptr(X), adj.data_ptr(),
                                                Z.data_ptr(), self._stream()), 'twog_gcn_fused_fwd')
        return X, adj, Z

    def gcn_embed1_bwd(self, x_human, n_nodes, ab, mean_invstd, w1, de1):
        ptr, fstride, nf = self._geo(x_human)
        dev = x_human.device
        nblk = max(1, min(2048, (nf * n_nodes + 31) // 32))  # 4 waves per block, >= 8 rows per wave
        partials = torch.empty(nblk * (320 + 8 * n_nodes), dtype=torch.float32, device=dev)
        dw1 = torch.empty(64, 4, dtype=torch.float32, device=dev)
        db1 = torch.empty(64, dtype=torch.float32, device=dev)
        dgamma = torch.empty(4 * n_nodes, dtype=torch.float32, device=dev)
        dbeta = torch.empty(4 * n_nodes, dtype=torch.float32, device=dev)
        self._check(self.lib.twog_gcn_embed1_bwd(ptr, fstride, nf, n_nodes, ab.data_ptr(), mean_invstd.data_ptr(),
                                                 w1.data_ptr(), de1.data_ptr(), partials.data_ptr(), nblk,
                                                 dw1.data_ptr(), db1.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                 self._stream()), 'twog_gcn_embed1_bwd')
        return dw1, db1, dgamma, dbeta

    def gcn_attn_fwd(self, qk, x, n_frames, n_nodes):
        s = torch.empty(n_frames, n_nodes, n_nodes, dtype=torch.float32, device=qk.device)
        z = torch.empty(n_frames * n_nodes, 64, dtype=torch.float32, device=qk.device)
        self._check(self.lib.twog_gcn_attn_fwd(qk.data_ptr(), x.data_ptr(), n_frames, n_nodes, s.data_ptr(),
                                               z.data_ptr(), self._stream()), 'twog_gcn_attn_fwd')
        return s, z

    def gcn_attn2_fwd(self, x, md, n_frames, n_nodes):
        """Folded-projection adjacency attention on MFMA: returns (adj [nF,N,N], z [nF*N,64])."""
        s = torch.empty(n_frames, n_nodes, n_nodes, dtype=torch.float32, device=x.device)
        z = torch.empty(n_frames * n_nodes, 64, dtype=torch.float32, device=x.device)
        self._check(self.lib.twog_gcn_attn2_fwd(x.data_ptr(), md.data_ptr(), n_frames, n_nodes, s.data_ptr(),
                                                z.data_ptr(), self._stream()), 'twog_gcn_attn2_fwd')
        return s, z

    def gcn_attn2_bwd(self, x, md, s, dz, n_frames, n_nodes):
        """Returns (dx_att [nF*N,64], dmd [65,64] = gradient wrt (Mt | d))."""
        dx = torch.empty(n_frames * n_nodes, 64, dtype=torch.float32, device=x.device)
        nblk = self.lib.twog_gcn_attn2_bwd_blocks(n_frames)
        partials = torch.empty(nblk, 65 * 64, dtype=torch.float32, device=x.device)
        self._check(self.lib.twog_gcn_attn2_bwd(x.data_ptr(), md.data_ptr(), s.data_ptr(), dz.data_ptr(), n_frames,
                                                n_nodes, dx.data_ptr(), partials.data_ptr(), nblk, self._stream()),
                    'twog_gcn_attn2_bwd')
        return dx, self.colsum(partials).view(65, 64)

    def gcn_attn_bwd(self, qk, x, s, dz, n_frames, n_nodes):
        dx = torch.empty(n_frames * n_nodes, 64, dtype=torch.float32, device=qk.device)
        dqk = torch.empty(n_frames * n_nodes, 256, dtype=torch.float32, device=qk.device)
        self._check(self.lib.twog_gcn_attn_bwd(qk.data_ptr(), x.data_ptr(), s.data_ptr(), dz.data_ptr(), n_frames,
                                               n_nodes, dx.data_ptr(), dqk.data_ptr(), self._stream()),
                    'twog_gcn_attn_bwd')
        return dx, dqk

    # ---------------------------------------------------------------- frame-level BiGRU
    def bigru_fwd(self, types, bs, T, h):
        """types: list of dicts {gi (bs,T,E,6h), w_hh_f, b_hh_f, w_hh_r, b_hh_r}. Returns [(out (bs,T,E,2h), save)]."""
        n = len(types)
        arr = (L.BiGru * n)()
        outs, keep = [], []
        for i, y in enumerate(types):
            gi = y['gi']
            E = gi.shape[2]
            dev = gi.device
            assert gi.is_contiguous() and gi.shape == (bs, T, E, 6 * h)
            out = torch.empty(bs, T, E, 2 * h, dtype=torch.float32, device=dev)
            save = torch.empty(2, bs, T, E, 4 * h, dtype=torch.float32, device=dev)
            tmp = torch.empty(2, bs * E, 3 * h, dtype=torch.float32, device=dev)
            zeros = self.zeros(bs * E, h, device=dev)
            keep += [tmp, zeros]
            a = arr[i]
            a.gi, a.w_hh_f, a.b_hh_f, a.w_hh_r, a.b_hh_r = (gi.data_ptr(), y['w_hh_f'].data_ptr(),
                                                             _ptr(y.get('b_hh_f')), y['w_hh_r'].data_ptr(),
                                                             _ptr(y.get('b_hh_r')))
            a.out, a.save, a.tmp_gh, a.zeros, a.E = out.data_ptr(), save.data_ptr(), tmp.data_ptr(), zeros.data_ptr(), E
            outs.append((out, save))
        self.last_bigru_persistent = self.bigru_persistent(arr, n, bs, h, dev)
        if self.last_bigru_persistent:
            # one persistent launch, W_hh slices resident in LDS (csrc/gru_persist.hip); sync words zeroed per call
            sync = self.zeros(1024, device=dev)
            keep.append(sync)
            rc = self.lib.twog_bigru_fwd_persistent(arr, n, bs, T, h, sync.data_ptr(), self._stream())
            if self._persistent_ok(rc, sync, dev, 'twog_bigru_fwd_persistent'):
                return outs
            self.last_bigru_persistent = False   # not resident / a wait ran out: the launch-per-step path re-runs the pass
        self._check(self.lib.twog_bigru_fwd(arr, n, bs, T, h, *self.chain_workspace(dev), self._stream()), 'twog_bigru_fwd')
        return outs

    # The persistent launches need every workgroup of their grid resident at once. Three lines of defence:
    #  (1) a device this process shares with other ranks of its own group (distributed.DataParallel finds out at
    #      construction: more ranks than devices) is listed in `shared_devices` and never gets a persistent launch;
    #  (2) the library asks the runtime's occupancy figure before launching and refuses a grid the device cannot hold
    #      (rc TWOG_PERSIST_NOT_RESIDENT);
    #  (3) a tenant nobody told us about (another process, another stream's long kernel, a CU mask): every wait inside the
    #      launch is bounded, a time-out sets the launch's error word and drains the grid; the host reads the word right
    #      after the launch (one 4-byte read-back; the stream is drained at that point, which costs the pipeline about one
    #      launch latency) and re-runs the pass on the launch-per-step path -- same buffers, written in place. A device on
    #      which that happened gets no persistent launches for the next PERSISTENT_BACKOFF calls.
    shared_devices = set()       # device indices (per process): ranks of one group share them
    persistent_fallbacks = 0     # passes re-run on the launch-per-step path after a persistent launch gave up (tests)
    persistent_refused = 0       # persistent launches the occupancy check refused
    PERSISTENT_BACKOFF = 64
    _backoff = {}                # device index -> calls left without persistent launches
    _refused = set()             # (launch, device index, shape ...) the occupancy check refused: not asked again

    @staticmethod
    def _dev_index(dev):
        return dev.index if dev.index is not None else torch.cuda.current_device()

    def persistent_allowed(self, dev):
        """False on a shared device, and for a while after a persistent launch on this device gave up."""
        i = self._dev_index(dev)
        if i in HipKernels.shared_devices:
            return False
        left = HipKernels._backoff.get(i, 0)
        if left > 0:
            HipKernels._backoff[i] = left - 1
            return False
        return True

    # When is the error word read? Reading it right after the launch drains the stream while the host waits, and the host
    # then has to refill the queue launch by launch (measured at 8 clips: ~0.4 ms of the 21.5 ms step per read-back). So:
    # the first PERSIST_SYNC_CALLS persistent launches on a device -- and every launch for PERSISTENT_BACKOFF calls after
    # a failure -- are checked AT ONCE and recovered transparently (the pass is re-run per step before anything consumes
    # its outputs). After that many clean launches the device is evidently ours: the word is copied to pinned host memory
    # behind the launch (asynchronously) and read at the END of the forward / backward pass (verify_persistent, called by
    # ops.tggcn_forward / tggcn_backward), when the copy has long landed. A failure found that late -- a tenant that
    # arrived in mid-training -- cannot be repaired behind the caller's back (consumers have run on incomplete outputs):
    # it raises, loudly, with the process and the context alive, and the next calls are checked at once again.
    # TWOG_PERSIST_CHECK=sync: always at once; =lazy: always at the end of the pass.
    PERSIST_SYNC_CALLS = 8
    _clean = {}       # device index -> persistent launches checked at once that completed
    _lazy = {}        # device index -> [(event, pinned int32 tensor, what)]
    persistent_late_failures = 0

    def _persistent_ok(self, rc, sync, dev, what, err_index=128):
        """True if the persistent launch ran to completion (or will be verified at the end of the pass). False -> the
        caller re-runs the pass on the launch-per-step path."""
        if rc == L.PERSIST_NOT_RESIDENT:
            HipKernels.persistent_refused += 1
            return False
        self._check(rc, what)
        i = self._dev_index(dev)
        mode = os.environ.get('TWOG_PERSIST_CHECK', 'auto')
        word = sync.view(torch.int32)[err_index:err_index + 1]
        if mode == 'lazy' or (mode != 'sync' and HipKernels._clean.get(i, 0) >= self.PERSIST_SYNC_CALLS):
            host, ev = self._lazy_slot(i)
            host.copy_(word, non_blocking=True)
            ev.record()
            HipKernels._lazy.setdefault(i, []).append((ev, host, what))
            return True
        if int(word.item()) == 0:   # (drains the stream: see above)
            HipKernels._clean[i] = HipKernels._clean.get(i, 0) + 1
            return True
        HipKernels.persistent_fallbacks += 1
        HipKernels._backoff[i] = self.PERSISTENT_BACKOFF
        HipKernels._clean[i] = 0
        return False

    _slots = {}   # device index -> ([pinned int32 [1] tensors], [events], next): a ring, so that a training step allocates nothing

    def _lazy_slot(self, i, n=32):
        ring = HipKernels._slots.get(i)
        if ring is None:
            pinned = torch.zeros(n, dtype=torch.int32).pin_memory()
            ring = [[pinned[k:k + 1] for k in range(n)], [torch.cuda.Event() for _ in range(n)], 0]
            HipKernels._slots[i] = ring
        k = ring[2]
        ring[2] = (k + 1) % n
        return ring[0][k], ring[1][k]

    def verify_persistent(self, dev=None):
        """End of a forward / backward pass: every persistent launch of the pass whose error word was left for later must
        have completed. Raises RuntimeError otherwise (see above)."""
        i = self._dev_index(dev if dev is not None else torch.device('cuda', torch.cuda.current_device()))
        pending = HipKernels._lazy.pop(i, None)
        if not pending:
            return
        failed = []
        for ev, host, what in pending:
            ev.synchronize()
            if int(host[0]) != 0:
                failed.append(what)
        if failed:
            HipKernels.persistent_late_failures += 1
            HipKernels._backoff[i] = self.PERSISTENT_BACKOFF
            HipKernels._clean[i] = 0
            raise RuntimeError(f'{", ".join(failed)}: a persistent launch could not keep its grid resident (another tenant '
                               'is holding compute units of this GPU) and gave up; the results of this pass are incomplete. '
                               'Repeat the step: the next calls run the launch-per-step path and persistent launches are '
                               're-admitted one checked launch at a time (TWOG_BIGRU_PERSIST=0 TWOG_SEG_PERSIST=0 switch '
                               'them off for good).')

    def guard_persistent(self, dev, outs):
        """End of a forward pass that a backward pass will follow: instead of waiting for the pass's deferred error words
        (verify_persistent drains the stream, and the backward pass is then enqueued into an idle device: ~1 ms per 8-clip
        step), ONE small launch behind them overwrites `outs` with NaN if any is set. The words stay pending: the check at
        the end of the backward pass (or at the start of the next forward) raises. True when nothing is pending or the guard
        was enqueued; False -> the caller verifies at once."""
        i = self._dev_index(dev if dev is not None else torch.device('cuda', torch.cuda.current_device()))
        pending = HipKernels._lazy.get(i)
        if not pending:
            return True
        outs = [o for o in outs if o is not None and o.numel()]
        if (len(pending) > L.GUARD_MAX or len(outs) > L.GUARD_MAX or os.environ.get('TWOG_PERSIST_GUARD', '1') == '0'
                or any(o.dtype != torch.float32 or not o.is_contiguous() for o in outs)):
            return False
        g = L.Guard()
        for k, (_, host, _) in enumerate(pending):
            g.words[k] = host.data_ptr()
        for k, o in enumerate(outs):
            g.out[k], g.n[k] = o.data_ptr(), o.numel()
        g.n_words, g.n_out = len(pending), len(outs)
        self._check(self.lib.twog_guard_outputs(C.byref(g), self._stream()), 'twog_guard_outputs')
        return True

    def bigru_persistent(self, arr, n, bs, h, dev=None):
        """True when the frame-level recurrence runs as the persistent launch: where the library serves the shape and
        rates it the faster path (small batches: at most one 16-row tile per wave). TWOG_BIGRU_PERSIST=1: wherever it is
        served; =0: never."""
        mode = os.environ.get('TWOG_BIGRU_PERSIST', 'auto')
        if mode == '0' or int(self.lib.twog_bigru_persistent_supported(arr, n, bs, h)) < (1 if mode == '1' else 2):
            return False
        return self.persistent_allowed(dev if dev is not None else torch.device('cuda', torch.cuda.current_device()))

    def bigru_bwd_would_persist(self, Es, bs, h):
        """Whether bigru_bwd takes the persistent launch for entity counts Es at this batch (asked before the operands exist:
        the data-parallel backward pass orders its first gradient all-reduce around that launch)."""
        i = self._dev_index(torch.device('cuda', torch.cuda.current_device()))
        # the same conditions bigru_bwd applies -- a shared device, the back-off window after a launch that gave up (read, not
        # consumed: persistent_allowed() counts it down when the launch is really attempted) and a grid the occupancy check
        # refused before on this device (the caller orders its first all-reduce and its side stream around the answer)
        if (os.environ.get('TWOG_BIGRU_PERSIST', 'auto') == '0' or i in HipKernels.shared_devices
                or HipKernels._backoff.get(i, 0) > 0 or ('bigru_bwd', i, tuple(Es), bs, h) in HipKernels._refused):
            return False
        arr = (L.BiGruBwd * len(Es))()
        for a, E in zip(arr, Es):
            a.E = E
        return int(self.lib.twog_bigru_bwd_persistent_supported(arr, len(Es), bs, h)) >= 2

    def bigru_bwd(self, types, bs, T, h, allow_persistent=True):
        """types: list of dicts {d_out, save, out, w_hh_f, w_hh_r}. Returns [(d_gi, d_gh)] each (bs,T,E,6h)."""
        n = len(types)
        arr = (L.BiGruBwd * n)()
        outs, keep = [], []
        for i, y in enumerate(types):
            d_out = y['d_out']
            E = d_out.shape[2]
            dev = d_out.device
            assert d_out.is_contiguous() and d_out.shape == (bs, T, E, 2 * h)
            d_gi = torch.empty(bs, T, E, 6 * h, dtype=torch.float32, device=dev)
            d_gh = torch.empty(bs, T, E, 6 * h, dtype=torch.float32, device=dev)
            carry = torch.empty(2, bs * E, h, dtype=torch.float32, device=dev)
            keep.append(carry)
            a = arr[i]
            a.d_out, a.save, a.out = d_out.data_ptr(), y['save'].data_ptr(), y['out'].data_ptr()
            a.w_hh_f, a.w_hh_r = y['w_hh_f'].data_ptr(), y['w_hh_r'].data_ptr()
            a.d_gi, a.d_gh, a.carry, a.E = d_gi.data_ptr(), d_gh.data_ptr(), carry.data_ptr(), E
            outs.append((d_gi, d_gh))
        mode = os.environ.get('TWOG_BIGRU_PERSIST', 'auto')
        self.last_bigru_bwd_persistent = (mode != '0' and allow_persistent
                                          and ('bigru_bwd', self._dev_index(dev), tuple(int(y['d_out'].shape[2]) for y in types), bs, h)
                                          not in HipKernels._refused
                                          and int(self.lib.twog_bigru_bwd_persistent_supported(arr, n, bs, h)) >= 2
                                          and self.persistent_allowed(dev))   # (last: it consumes a back-off credit)
        if self.last_bigru_bwd_persistent:   # small batches: one persistent launch (csrc/gru_persist.hip)
            sync = self.zeros(1024, device=dev)
            keep.append(sync)
            rc = self.lib.twog_bigru_bwd_persistent(arr, n, bs, T, h, sync.data_ptr(), self._stream())
            if rc == L.PERSIST_NOT_RESIDENT:
                HipKernels._refused.add(('bigru_bwd', self._dev_index(dev), tuple(int(y['d_out'].shape[2]) for y in types), bs, h))
            if self._persistent_ok(rc, sync, dev, 'twog_bigru_bwd_persistent'):
                return outs
            self.last_bigru_bwd_persistent = False
        self._check(self.lib.twog_bigru_bwd(arr, n, bs, T, h, *self.chain_workspace(dev), self._stream()), 'twog_bigru_bwd')
        return outs

    # ---------------------------------------------------------------- single GRU gate steps (general segment loop)
    @staticmethod
    def _u_fields(g, u):
        """u: None or a (bs, E) view of a (bs, T, E) gate tensor at one time step."""
        if u is None:
            g.u, g.u_ld_outer, g.u_ld_inner, g.u_inner = 0, 0, 0, 1
        else:
            g.u, g.u_ld_outer, g.u_ld_inner, g.u_inner = u.data_ptr(), u.stride(0), u.stride(1), u.shape[1]

    def gru_step_fwd(self, steps):
        """steps: dicts with gi, gi2 (or None), gh, h_prev (or None), h_out, save (row-strided views), u, rows, hidden."""
        arr = (L.GruStep * len(steps))()
        for g, d in zip(arr, steps):
            for k in ('gi', 'gi2', 'gh', 'h_prev', 'h_out', 'save'):
                setattr(g, k, rows_of(d.get(k)))
            self._u_fields(g, d.get('u'))
            g.rows, g.hidden = d['rows'], d['hidden']
        if self._tape is not None:
            self._tape.append((L.TAPE_GRU_STEP_FWD, len(steps), 0, arr))
            return
        self._check(self.lib.twog_gru_step_fwd(arr, len(steps), self._stream()), 'twog_gru_step_fwd')

    def gru_step_bwd(self, steps):
        """steps: dicts with dh, dh2 (or None), save, h_prev (or None), dgi, dgh, dh_prev, u, du (same view form as u)."""
        arr = (L.GruStepBwd * len(steps))()
        for g, d in zip(arr, steps):
            for k in ('dh', 'dh2', 'save', 'h_prev', 'dgi', 'dgh', 'dh_prev'):
                setattr(g, k, rows_of(d.get(k)))
            self._u_fields(g, d.get('u'))
            g.du = _ptr(d.get('du'))
            g.rows, g.hidden, g.dh_prev_accumulate = d['rows'], d['hidden'], int(d.get('dh_prev_accumulate', 0))
        if self._tape is not None:
            self._tape.append((L.TAPE_GRU_STEP_BWD, len(steps), 0, arr))
            return
        self._check(self.lib.twog_gru_step_bwd(arr, len(steps), self._stream()), 'twog_gru_step_bwd')

    # ---------------------------------------------------------------- entity attention
    _ATTN_ROWS = ['feat_h', 'feat_o', 'msg_hh', 'msg_ho', 'msg_oh', 'msg_oo', 'msg_so', 'msg_sh', 'out_hh', 'out_oh',
                  'out_sh', 'out_ho', 'out_so', 'out_oo']

    def _fill_attn(self, a, d):
        for k in self._ATTN_ROWS:
            setattr(a, k, rows_of(d.get(k)))
        a.obj_mask = _ptr(d.get('obj_mask'))
        a.att = _ptr(d.get('att'))
        a.n_inst, a.inst_per_clip, a.H, a.O = d['n_inst'], d['inst_per_clip'], d['H'], d['O']
        a.D, a.hidden, a.scale, a.recv_mask_ho = d['D'], d['hidden'], float(d['scale']), int(d['recv_mask_ho'])

    def attn_fwd(self, descs):
        n = len(descs)
        arr = (L.Attn * n)()
        for i, d in enumerate(descs):
            self._fill_attn(arr[i], d)
        self._check(self.lib.twog_attn_fwd(arr, n, self._stream()), 'twog_attn_fwd')

    def attn_bwd(self, descs):
        n = len(descs)
        arr = (L.AttnBwd * n)()
        for i, d in enumerate(descs):
            self._fill_attn(arr[i].f, d['f'])
            for k in ['dout_hh', 'dout_oh', 'dout_sh', 'dout_ho', 'dout_so', 'dout_oo', 'dmsg_hh', 'dmsg_ho',
                      'dmsg_oh', 'dmsg_oo', 'dmsg_so', 'dmsg_sh', 'dfeat_h', 'dfeat_o']:
                setattr(arr[i], k, rows_of(d.get(k)))
            arr[i].dw_extra = _ptr(d.get('dw_extra'))
            arr[i].dfeat_accumulate = int(d.get('dfeat_accumulate', 0))
            arr[i].relu_mask_dmsg = int(d.get('relu_mask_dmsg', 0))
        self._check(self.lib.twog_attn_bwd(arr, n, self._stream()), 'twog_attn_bwd')

    # ---------------------------------------------------------------- segment-level recurrence
    @staticmethod
    def _seg_dims(p):
        nsh = int(p['rel_hh']) + int(p['rel_ho'])
        nso = int(p['rel_oh']) + int(p['rel_oo'])
        nmh = int(p['rel_hh']) + int(p['rel_oh'])
        nmo = int(p['rel_ho']) + int(p['rel_oo'])
        return nsh, nso, nmh, nmo

    def _fill_seg(self, s, p, bufs):
        s.bs, s.T, s.H, s.O, s.hidden = p['bs'], p['T'], p['H'], p['O'], p['hidden']
        s.msg_segment = int(p['msg_segment'])
        s.rel_hh, s.rel_ho, s.rel_oh, s.rel_oo = (int(p['rel_hh']), int(p['rel_ho']), int(p['rel_oh']),
                                                   int(p['rel_oo']))
        s.att_scale = float(p['att_scale'])
        s.gi_h, s.gi_o, s.u_h, s.u_o = _ptr(p['gi_h']), _ptr(p['gi_o']), _ptr(p['u_h']), _ptr(p['u_o'])
        s.obj_mask = _ptr(p['obj_mask'])
        for d in range(2):
            s.w_hh_h[d], s.b_hh_h[d] = _ptr(p['w_hh_h'][d]), _ptr(p['b_hh_h'][d])
            s.w_hh_o[d], s.b_hh_o[d] = _ptr(p['w_hh_o'][d]), _ptr(p['b_hh_o'][d])
            s.w_ihm_h[d], s.w_ihm_o[d] = _ptr(p['w_ihm_h'][d]), _ptr(p['w_ihm_o'][d])
        s.ld_ih_h, s.ld_ih_o = p['ld_ih_h'], p['ld_ih_o']
        s.w_smsg_h, s.b_smsg_h = _ptr(p.get('w_smsg_h')), _ptr(p.get('b_smsg_h'))
        s.w_smsg_o, s.b_smsg_o = _ptr(p.get('w_smsg_o')), _ptr(p.get('b_smsg_o'))
        for k in ['hs_h', 'hs_o', 'save_h', 'save_o', 'msrc_h', 'msrc_o', 'mg_h', 'mg_o', 'att', 'tmp_gim_h',
                  'tmp_gim_o', 'tmp_gh_h', 'tmp_gh_o', 'zeros']:
            setattr(s, k, _ptr(bufs[k]))

    def segrnn_fwd(self, p):
        """p: parameter/input dict (see ops.SegmentRecurrence). Allocates and returns the state/saved buffers."""
        bs, T, H, O, h = p['bs'], p['T'], p['H'], p['O'], p['hidden']
        dev = p['gi_h'].device if p['gi_h'] is not None else p['gi_o'].device
        nsh, nso, nmh, nmo = self._seg_dims(p)
        natt = H * H + 2 * H * O + O * O

        def e(*shape):
            return torch.empty(*[max(int(x), 0) for x in shape], dtype=torch.float32, device=dev)

        bufs = dict(hs_h=e(bs, T, H, 2 * h), hs_o=e(bs, T, O, 2 * h), save_h=e(2, bs, T, H, 4 * h),
                    save_o=e(2, bs, T, O, 4 * h), msrc_h=e(2, bs, T, H, nsh * h), msrc_o=e(2, bs, T, O, nso * h),
                    mg_h=e(2, bs, T, H, nmh * h), mg_o=e(2, bs, T, O, nmo * h), att=e(2, T, bs, natt),
                    tmp_gim_h=e(2, bs * H, 3 * h), tmp_gim_o=e(2, bs * O, 3 * h), tmp_gh_h=e(2, bs * H, 3 * h),
                    tmp_gh_o=e(2, bs * O, 3 * h),
                    zeros=self.zeros(bs * max(H, O, 1), h, device=dev))
        s = L.SegRnn()
        self._fill_seg(s, p, bufs)
        mode = os.environ.get('TWOG_SEG_PERSIST', 'auto')
        self.last_segrnn_persistent = (mode != '0' and int(self.lib.twog_segrnn_persistent_supported(C.byref(s))) >= 2
                                       and self.persistent_allowed(dev))   # (last: it consumes a back-off credit)
        if self.last_segrnn_persistent:   # small batches: the whole recurrence in one launch (csrc/seg_persist.hip)
            n_sync = int(self.lib.twog_segrnn_persistent_sync_bytes()) // 4
            sync = self.zeros(n_sync, device=dev)
            rc = self.lib.twog_segrnn_fwd_persistent(C.byref(s), sync.data_ptr(), self._stream())
            if self._persistent_ok(rc, sync, dev, 'twog_segrnn_fwd_persistent', err_index=n_sync - 32):
                return bufs
            self.last_segrnn_persistent = False
        self._check(self.lib.twog_segrnn_fwd(C.byref(s), *self.chain_workspace(dev), self._stream()), 'twog_segrnn_fwd')
        return bufs

    def segrnn_bwd(self, p, bufs, d_hs_h, d_hs_o):
        bs, T, H, O, h = p['bs'], p['T'], p['H'], p['O'], p['hidden']
        dev = bufs['hs_h'].device
        nsh, nso, nmh, nmo = self._seg_dims(p)

        def e(*shape):
            return torch.empty(*[max(int(x), 0) for x in shape], dtype=torch.float32, device=dev)

        d_u_h, d_u_o = self.zeros_many([(bs, T, max(H, 0)), (bs, T, max(O, 0))], dev)
        out = dict(d_gi_h=e(bs, T, H, 6 * h), d_gi_o=e(bs, T, O, 6 * h), d_gh_h=e(bs, T, H, 6 * h),
                   d_gh_o=e(bs, T, O, 6 * h),
                   d_u_h=d_u_h, d_u_o=d_u_o,
                   d_pre_h=e(2, bs, T, H, nsh * h), d_pre_o=e(2, bs, T, O, nso * h))
        scratch = dict(carry_h=e(2, bs * H, h), carry_o=e(2, bs * O, h), tmp_dmg_h=e(2, bs * H, nmh * h),
                       tmp_dmg_o=e(2, bs * O, nmo * h), trash=e(bs * max(H, O, 1), h),
                       du_part_h=e(2, T, 16, bs * H), du_part_o=e(2, T, 16, bs * O))
        s = L.SegRnn()
        self._fill_seg(s, p, bufs)
        b = L.SegRnnBwd()
        b.d_hs_h, b.d_hs_o = _ptr(d_hs_h), _ptr(d_hs_o)
        for k, v in list(out.items()) + list(scratch.items()):
            setattr(b, k, _ptr(v))
        mode = os.environ.get('TWOG_SEG_PERSIST', 'auto')
        self.last_segrnn_bwd_persistent = (mode != '0' and int(self.lib.twog_segrnn_persistent_supported(C.byref(s))) >= 2
                                           and self.persistent_allowed(dev))
        if self.last_segrnn_bwd_persistent:   # small batches: backward through time in one launch (csrc/seg_persist.hip)
            n_scr = int(self.lib.twog_segrnn_bwd_persistent_scratch_bytes(C.byref(s)))
            scr = self.workspace(n_scr, dev, 'segp_bwd')
            n_sync = int(self.lib.twog_segrnn_persistent_sync_bytes()) // 4
            sync = self.zeros(n_sync, device=dev)
            rc = self.lib.twog_segrnn_bwd_persistent(C.byref(s), C.byref(b), scr.data_ptr(), scr.numel() * 4, sync.data_ptr(),
                                                     self._stream())
            if self._persistent_ok(rc, sync, dev, 'twog_segrnn_bwd_persistent', err_index=n_sync - 32):
                return out
            self.last_segrnn_bwd_persistent = False
        self._check(self.lib.twog_segrnn_bwd(C.byref(s), C.byref(b), *self.chain_workspace(dev), self._stream()), 'twog_segrnn_bwd')
        return out

    def graph_cache_stats(self):
        """(captured loops, hash-bucket hits resolved by the descriptor compare) of the library's hipGraph cache."""
        n, c = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.twog_graph_cache_stats(C.byref(n), C.byref(c)), 'twog_graph_cache_stats')
        return n.value, c.value

    # ---------------------------------------------------------------- gates
    def _fill_gate(self, g, d):
        g.x = rows_of(d['x'])
        for i, c in enumerate(d['seg_col']):
            g.seg_col[i] = int(c)
        g.n_seg, g.hidden = len(d['seg_col']), d['hidden']
        g.w, g.b, g.noise = _ptr(d['w']), _ptr(d.get('b')), _ptr(d.get('noise'))
        g.hard, g.soft, g.p_save = _ptr(d['hard']), _ptr(d['soft']), _ptr(d['p_save'])
        g.bs, g.T, g.E = d['bs'], d['T'], d['E']
        g.noise_entities, g.noise_offset = int(d.get('noise_entities', 0)), int(d.get('noise_offset', 0))
        g.force_last, g.threshold = int(d['force_last']), float(d['threshold'])

    def gate_fwd(self, d):
        """d: x (rows (bs*T*E), W) view, seg_col, hidden, w, b, noise, bs, T, E, ... Fills d['hard'|'soft'|'p_save']."""
        dev = d['x'].device
        for k in ('hard', 'soft', 'p_save'):
            d[k] = torch.empty(d['bs'], d['T'], d['E'], dtype=torch.float32, device=dev)
        g = L.Gate()
        self._fill_gate(g, d)
        self._check(self.lib.twog_gate_fwd(C.byref(g), self._stream()), 'twog_gate_fwd')
        return d['hard'], d['soft']

    def gate_bwd(self, d, d_hard, d_soft, st_mask):
        g = L.Gate()
        self._fill_gate(g, d)
        dlogit = torch.empty(d['bs'] * d['T'] * d['E'], dtype=torch.float32, device=d['x'].device)
        self._check(self.lib.twog_gate_bwd(C.byref(g), _ptr(d_hard), _ptr(d_soft), _ptr(st_mask), dlogit.data_ptr(),
                                           self._stream()), 'twog_gate_bwd')
        return dlogit

    def rank1_update(self, dst, s, v):
        """dst[r][c] += s[r] * v[c] on a row-strided view."""
        self._check(self.lib.twog_rank1_update(rows_of(dst), s.data_ptr(), v.data_ptr(), n_rows(dst), dst.shape[-1],
                                               self._stream()), 'twog_rank1_update')

    def colsum(self, x, rowscale=None, out=None, accumulate=False):
        rows, cols = n_rows(x), x.shape[-1]
        if out is None:
            out = torch.empty(cols, dtype=torch.float32, device=x.device)
            accumulate = False
        # enough row slices to fill the chip (~1024 workgroups over column blocks x row slices), >= 64 rows per slice
        nblk = max(1, min(rows // 64, max(1, 1024 // ((cols + 255) // 256))))
        partials = self.workspace(nblk * cols * 4, x.device, 'colsum')
        self._check(self.lib.twog_colsum(rows_of(x), _ptr(rowscale), rows, cols, out.data_ptr(), int(accumulate),
                                         partials.data_ptr(), nblk, self._stream()), 'twog_colsum')
        return out

    def colsum_many(self, ops):
        """ops: (x, rowscale or None, out [cols], accumulate) -- several column sums in one pair of launches (twog_colsum_n);
        each with the arithmetic of its own colsum() call."""
        if not ops:
            return
        arr = (L.ColSum * len(ops))()
        for a, (x, rs, out, acc) in zip(arr, ops):
            assert out.is_contiguous() and out.numel() == x.shape[-1]
            a.x, a.rowscale, a.out = rows_of(x), _ptr(rs), out.data_ptr()
            a.rows, a.cols, a.accumulate = n_rows(x), x.shape[-1], int(bool(acc))
        dev = ops[0][0].device
        need = int(self.lib.twog_colsum_n_partial_floats(arr, len(ops)))
        partials = self.workspace(max(need, 1) * 4, dev, 'colsum')
        self._check(self.lib.twog_colsum_n(arr, len(ops), partials.data_ptr(), partials.numel(), self._stream()), 'twog_colsum_n')

    def filter_fwd(self, soft, threshold):
        bs, T, E = soft.shape
        hard, gmask = torch.empty_like(soft), torch.empty_like(soft)
        self._check(self.lib.twog_filter_fwd(soft.data_ptr(), hard.data_ptr(), gmask.data_ptr(), bs, T, E,
                                             float(threshold), self._stream()), 'twog_filter_fwd')
        return hard, gmask

    # ---------------------------------------------------------------- reorder / heads / elementwise
    def reorder_fwd(self, hx, gate):
        bs, T, E, cols = hx.shape
        assert hx.is_contiguous() and gate.is_contiguous() and gate.shape == (bs, T, E)
        out = torch.empty_like(hx)
        self._check(self.lib.twog_reorder_fwd(hx.data_ptr(), gate.data_ptr(), out.data_ptr(), bs, T, E, cols,
                                              self._stream()), 'twog_reorder_fwd')
        return out

    def reorder_bwd(self, dout, gate):
        bs, T, E, cols = dout.shape
        assert dout.is_contiguous()
        dhx = torch.empty_like(dout)
        self._check(self.lib.twog_reorder_bwd(dout.data_ptr(), gate.data_ptr(), dhx.data_ptr(), bs, T, E, cols,
                                              self._stream()), 'twog_reorder_bwd')
        return dhx

    def logsoftmax_permute_fwd(self, logits, bs, T, E, Cn):
        out = torch.empty(bs, Cn, T, E, dtype=torch.float32, device=logits.device)
        self._check(self.lib.twog_logsoftmax_permute_fwd(logits.data_ptr(), out.data_ptr(), bs, T, E, Cn,
                                                         self._stream()), 'twog_logsoftmax_permute_fwd')
        return out

    def logsoftmax_permute_bwd(self, out, dout):
        bs, Cn, T, E = out.shape
        assert out.is_contiguous() and dout.is_contiguous()
        dlogits = torch.empty(bs * T * E, Cn, dtype=torch.float32, device=out.device)
        self._check(self.lib.twog_logsoftmax_permute_bwd(out.data_ptr(), dout.data_ptr(), dlogits.data_ptr(), bs, T,
                                                         E, Cn, self._stream()), 'twog_logsoftmax_permute_bwd')
        return dlogits

    def relu_bwd(self, dy, y, dx=None):
        """dx = dy * (y > 0) on row-strided views (dx may alias dy)."""
        if dx is None:
            dx = torch.empty(n_rows(dy), dy.shape[-1], dtype=torch.float32, device=dy.device)
        self._check(self.lib.twog_relu_bwd(rows_of(dy), rows_of(y), rows_of(dx), n_rows(dy), dy.shape[-1],
                                           self._stream()), 'twog_relu_bwd')
        return dx

    def add_rows(self, src, dst):
        self._check(self.lib.twog_add_rows(rows_of(src), rows_of(dst), n_rows(src), src.shape[-1], self._stream()),
                    'twog_add_rows')

    # ---------------------------------------------------------------- sender-side projection glue
    def ssp_fwd(self, gi, ph, ps, att, mask, n_inst, inst_per_clip, H, O, att_off):
        """gi (n_inst*O, cols) += mask * (sum_h att[k,h] ph[(inst,h)] + ps[inst]); ph (n_inst*H, cols), ps (n_inst, cols)."""
        cols = gi.shape[-1]
        assert gi.is_contiguous() and (ph is None or ph.is_contiguous()) and (ps is None or ps.is_contiguous())
        natt = att.shape[-1] if att is not None else 0
        self._check(self.lib.twog_ssp_fwd(gi.data_ptr(), _ptr(ph), _ptr(ps), _ptr(att), _ptr(mask), n_inst, inst_per_clip,
                                          H, O, cols, natt, att_off, self._stream()), 'twog_ssp_fwd')

    def ssp_gather(self, dgi, att, att_ld_clip, att_ld_frame, att_off, n_inst, inst_per_clip, H, O):
        """dgi: (n_inst*O, cols) view with unit column stride (row stride free) -> qh (n_inst*H, cols) =
        sum_k att(inst)[att_off + k*H + h] * dgi[(inst, k)]."""
        assert dgi.dim() == 2 and dgi.stride(1) == 1
        cols = dgi.shape[1]
        qh = torch.empty(n_inst * H, cols, dtype=torch.float32, device=dgi.device)
        self._check(self.lib.twog_ssp_gather(dgi.data_ptr(), dgi.stride(0), att.data_ptr(), att_ld_clip, att_ld_frame,
                                             att_off, qh.data_ptr(), n_inst, inst_per_clip, H, O, cols, self._stream()),
                    'twog_ssp_gather')
        return qh

    def ssp_bwd(self, dgi, ph, att, mask, n_inst, inst_per_clip, H, O, att_off, want_qs, dw=None):
        """Returns (qh (n_inst*H, cols) or None, qs (n_inst, cols) or None); fills dw[:, att_off : att_off + O*H] if given."""
        cols = dgi.shape[-1]
        assert dgi.is_contiguous() and (ph is None or ph.is_contiguous())
        qh = torch.empty(n_inst * H, cols, dtype=torch.float32, device=dgi.device) if ph is not None else None
        qs = torch.empty(n_inst, cols, dtype=torch.float32, device=dgi.device) if want_qs else None
        natt = att.shape[-1] if att is not None else 0
        self._check(self.lib.twog_ssp_bwd(dgi.data_ptr(), _ptr(ph), _ptr(att), _ptr(mask), _ptr(qh), _ptr(qs), _ptr(dw),
                                          n_inst, inst_per_clip, H, O, cols, natt, att_off, self._stream()),
                    'twog_ssp_bwd')
        return qh, qs

    # ---------------------------------------------------------------- general single-relation message passing
    REL_SUM, REL_DOT, REL_ADDITIVE, REL_DISTANCE, REL_MEAN = (L.REL_SUM, L.REL_DOT, L.REL_ADDITIVE, L.REL_DISTANCE,
                                                              L.REL_MEAN)
    REL_MSG_SENDER, REL_MSG_PAIR = L.REL_MSG_SENDER, L.REL_MSG_PAIR

    def _fill_relation(self, a, d):
        for k in ('q', 'k', 'msg', 'p_r', 'p_s', 'out'):
            setattr(a, k, rows_of(d.get(k)))
        a.a_r, a.c_s = _ptr(d.get('a_r')), _ptr(d.get('c_s'))
        dist = d.get('dist')   # (n_inst, R, S) view, any strides
        if dist is not None:
            assert dist.dim() == 3 and dist.dtype == torch.float32
            a.dist, (a.dist_ld_inst, a.dist_ld_r, a.dist_ld_s) = dist.data_ptr(), dist.stride()
        else:
            a.dist, a.dist_ld_inst, a.dist_ld_r, a.dist_ld_s = 0, 0, 0, 0
        a.send_mask, a.recv_mask, a.att = _ptr(d.get('send_mask')), _ptr(d.get('recv_mask')), _ptr(d.get('att'))
        a.scale, a.score_bias = float(d.get('scale', 1.0)), _ptr(d.get('score_bias'))
        a.score_mode, a.msg_mode = int(d['score_mode']), int(d['msg_mode'])
        a.relu_scores, a.exclude_self = int(d.get('relu_scores', 0)), int(d.get('exclude_self', 0))
        a.n_inst, a.inst_per_clip, a.R, a.S = d['n_inst'], d['inst_per_clip'], d['R'], d['S']
        a.D, a.hidden = int(d.get('D', 0)), d['hidden']

    def relation_fwd(self, d):
        a = L.Relation()
        self._fill_relation(a, d)
        self._check(self.lib.twog_relation_fwd(C.byref(a), self._stream()), 'twog_relation_fwd')

    def relation_fwd_many(self, ds):
        """Several relations in one call (a few launches of up to 8 descriptors): the general segment loop's level."""
        arr = (L.Relation * len(ds))()
        for a, d in zip(arr, ds):
            self._fill_relation(a, d)
        if self._tape is not None:
            self._tape.append((L.TAPE_RELATION_FWD, len(ds), 0, arr))
            return
        self._check(self.lib.twog_relation_fwd_n(arr, len(ds), self._stream()), 'twog_relation_fwd_n')

    def relation_bwd_many(self, ds):
        """The descriptors must not accumulate (dq / dk) into the same rows: they run concurrently."""
        arr = (L.RelationBwd * len(ds))()
        for b, d in zip(arr, ds):
            self._fill_relation_bwd(b, d)
        if self._tape is not None:
            self._tape.append((L.TAPE_RELATION_BWD, len(ds), 0, arr))
            return
        self._check(self.lib.twog_relation_bwd_n(arr, len(ds), self._stream()), 'twog_relation_bwd_n')

    _ROWOP = {'relu_bwd': 0, 'add': 1, 'rank1': 2}   # TWOG_ROWOP_*

    def rowops(self, ops):
        """ops: ('relu_bwd', dy, y, dx) | ('add', src, dst) | ('rank1', dst, s, v), row-strided views; ONE launch per 16
        operations. No two operations of a call may write the same rows."""
        if not ops:
            return
        arr = (L.RowOp * len(ops))()
        none = rows_of(None)
        for o, op in zip(arr, ops):
            o.kind = self._ROWOP[op[0]]
            if op[0] == 'relu_bwd':
                _, dy, y, dst = op
                o.a, o.b = rows_of(dy), rows_of(y)
            elif op[0] == 'add':
                _, src, dst = op
                o.a, o.b = rows_of(src), none
            else:
                _, dst, sv, v = op
                assert sv.is_contiguous() and v.is_contiguous() and sv.numel() == n_rows(dst) and v.numel() == dst.shape[-1]
                o.a, o.b, o.s, o.v = none, none, sv.data_ptr(), v.data_ptr()
            o.dst, o.rows, o.cols = rows_of(dst), n_rows(dst), dst.shape[-1]
        if self._tape is not None:
            self._tape.append((L.TAPE_ROWOPS, len(ops), 0, arr))
            return
        self._check(self.lib.twog_rowops(arr, len(ops), self._stream()), 'twog_rowops')

    # ---------------------------------------------------------------- recorded steps of an affine loop (twog_tape_run)
    def tape_begin(self):
        """From here to tape_end(), gemm / gru_step_* / relation_*_many / rowops calls are RECORDED (descriptor arrays
        kept, nothing issued). Any other entry point called in between is a programming error the recording cannot see:
        the general segment loop (ops.segment_recurrence_general_*) uses only these six."""
        assert self._tape is None
        self._tape = []

    def tape_end(self):
        t, self._tape = self._tape, None
        return t

    @staticmethod
    def tape_predict(a, b, k):
        """The descriptor bytes twog_tape_run builds for step a + k: per 64-bit word a + k (b - a). For checking a third
        composed step against the rule before trusting it with the rest of the loop."""
        import numpy as np
        out = []
        for (ka, na, fa, da), (kb, nb, fb, db) in zip(a, b):
            wa = np.frombuffer(bytes(da), dtype=np.uint64)
            wb = np.frombuffer(bytes(db), dtype=np.uint64)
            out.append((wa + np.uint64(k) * (wb - wa)).tobytes())
        return out

    @staticmethod
    def tape_same_program(a, b):
        return len(a) == len(b) and all(x[:3] == y[:3] for x, y in zip(a, b))

    def tape_matches(self, a, b, c, k):
        """True if the composed step c is what the replay would run as step a + k."""
        if not (self.tape_same_program(a, b) and self.tape_same_program(a, c)):
            return False
        return all(bytes(e[3]) == w for e, w in zip(c, self.tape_predict(a, b, k)))

    def tape_run(self, a, b, k_begin, k_end, device):
        """Issues steps a + k, k_begin <= k < k_end, of the loop whose consecutive steps a and b were recorded."""
        assert self._tape is None and self.tape_same_program(a, b)
        n = len(a)
        ea, eb = (L.TapeEntry * n)(), (L.TapeEntry * n)()
        for arr, tape in ((ea, a), (eb, b)):
            for e, (kind, cnt, flags, desc) in zip(arr, tape):
                e.kind, e.n, e.flags, e.desc = kind, cnt, flags, C.addressof(desc)
        ws = self.workspace(320 << 20, device, 'splitk')
        self._check(self.lib.twog_tape_run(ea, eb, n, k_begin, k_end, ws.data_ptr(), ws.numel() * 4, self._stream()),
                    'twog_tape_run')

    def relation_bwd(self, d):
        """d: dict(f=<forward descriptor>, dout, dmsg | dp_r, dp_s, dq, dk, da_r, dc_s, dq_accumulate, ...)."""
        b = L.RelationBwd()
        self._fill_relation_bwd(b, d)
        self._check(self.lib.twog_relation_bwd(C.byref(b), self._stream()), 'twog_relation_bwd')

    def _fill_relation_bwd(self, b, d):
        self._fill_relation(b.f, d['f'])
        for k in ('dout', 'dmsg', 'dp_r', 'dp_s', 'dq', 'dk'):
            setattr(b, k, rows_of(d.get(k)))
        b.da_r, b.dc_s, b.dscore_sum = _ptr(d.get('da_r')), _ptr(d.get('dc_s')), _ptr(d.get('dscore_sum'))
        b.dq_accumulate, b.dk_accumulate = int(d.get('dq_accumulate', 0)), int(d.get('dk_accumulate', 0))
        b.relu_mask_dmsg = int(d.get('relu_mask_dmsg', 0))

    # ---------------------------------------------------------------- position features / rare gate strategies
    def pos_embed_fwd(self, out, bs, T, E, hidden, w=None, b=None, periodic=False, s=None, steps=None, divide=False):
        """out: (bs*T*E, hidden) row-strided view (a column block of the entity rows). Returns the scalars [bs*T*E]."""
        s_out = torch.empty(bs * T * E, dtype=torch.float32, device=out.device)
        self._check(self.lib.twog_pos_embed_fwd(_ptr(s), _ptr(steps), bs, T, E, int(divide), _ptr(w), _ptr(b),
                                                int(periodic), hidden, rows_of(out), s_out.data_ptr(), self._stream()),
                    'twog_pos_embed_fwd')
        return s_out

    def periodic_embed_bwd(self, dout, s):
        ds = torch.empty(n_rows(dout), dtype=torch.float32, device=dout.device)
        self._check(self.lib.twog_periodic_embed_bwd(rows_of(dout), s.data_ptr(), n_rows(dout), dout.shape[-1],
                                                     ds.data_ptr(), self._stream()), 'twog_periodic_embed_bwd')
        return ds

    def seglen_fwd(self, u, steps, divide):
        bs, T, E = u.shape
        assert u.is_contiguous()
        s = torch.empty(bs, T, E, dtype=torch.float32, device=u.device)
        self._check(self.lib.twog_seglen_fwd(u.data_ptr(), _ptr(steps), bs, T, E, int(divide), s.data_ptr(),
                                             self._stream()), 'twog_seglen_fwd')
        return s

    def seglen_bwd(self, u, steps, divide, ds, du):
        """du (bs, T, E) += d(segment lengths)/d(hard gates) applied to ds."""
        bs, T, E = u.shape
        assert ds.is_contiguous() and du.is_contiguous()
        self._check(self.lib.twog_seglen_bwd(u.data_ptr(), _ptr(steps), bs, T, E, int(divide), ds.data_ptr(),
                                             du.data_ptr(), self._stream()), 'twog_seglen_bwd')

    def mul(self, a, b, out=None, accumulate=False):
        assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
        if out is None:
            out, accumulate = torch.empty_like(a), False
        assert out.is_contiguous()
        self._check(self.lib.twog_mul(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), int(accumulate),
                                      self._stream()), 'twog_mul')
        return out

    def scale_rows(self, x, s):
        self._check(self.lib.twog_scale_rows(rows_of(x), s.data_ptr(), n_rows(x), x.shape[-1], self._stream()),
                    'twog_scale_rows')

    def adam_step(self, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
        self._check(self.lib.twog_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(),
                                            exp_avg_sq.data_ptr(), param.numel(), lr, beta1, beta2, eps, weight_decay,
                                            step, grad_scale, self._stream()), 'twog_adam_step')


    # ---------------------------------------------------------------- multi-task loss
    def _loss_terms(self, terms, dinputs=None):
        """terms: list of dict(kind 0 NLL / 1 BCE / 2 budget, input, target, weight, ignore). input/target contiguous."""
        arr = (L.Loss * len(terms))()
        for i, (t, a) in enumerate(zip(terms, arr)):
            x, y = t['input'], t['target']
            assert x.is_contiguous() and y.is_contiguous() and x.dtype == torch.float32
            a.kind = t['kind']
            if t['kind'] == 0:
                assert y.dtype == torch.int64 and x.dim() >= 2 and y.numel() * x.shape[1] == x.numel()
                a.n_classes, a.outer = x.shape[1], x.shape[0]
                a.inner = x.numel() // max(x.shape[0] * x.shape[1], 1)
            else:
                assert y.dtype == torch.float32 and y.numel() == x.numel()
                a.n_classes, a.outer, a.inner = 1, 1, x.numel()
            a.input, a.target = x.data_ptr(), y.data_ptr()
            a.dinput = None if dinputs is None or dinputs[i] is None else dinputs[i].data_ptr()
            a.weight, a.ignore_value = float(t['weight']), float(t['ignore'])
        return arr

    def multitask_loss_fwd(self, terms):
        """All loss terms in one launch. Returns (losses [n] fp32, stats [n][2] fp64 = sum, count)."""
        n, dev = len(terms), terms[0]['input'].device
        if n > L.LOSS_MAX_TERMS:
            raise ValueError(f'at most {L.LOSS_MAX_TERMS} loss terms per call')
        losses = torch.empty(n, dtype=torch.float32, device=dev)
        stats = torch.empty(n, 2, dtype=torch.float64, device=dev)
        partials = self.workspace(n * L.LOSS_BLOCKS * 2 * 8, dev, 'loss')
        self._check(self.lib.twog_multitask_loss_fwd(self._loss_terms(terms), n, partials.data_ptr(), stats.data_ptr(),
                                                     losses.data_ptr(), self._stream()), 'twog_multitask_loss_fwd')
        return losses, stats

    def multitask_loss_bwd(self, terms, stats, dlosses, need):
        """d(input) of every term with need[i] (others None), scaled by the upstream gradient dlosses [n]."""
        dins = [torch.empty_like(t['input']) if nd else None for t, nd in zip(terms, need)]
        self._check(self.lib.twog_multitask_loss_bwd(self._loss_terms(terms, dins), len(terms), stats.data_ptr(),
                                                     dlosses.contiguous().data_ptr(), self._stream()),
                    'twog_multitask_loss_bwd')
        return dins


    # ---------------------------------------------------------------- inference post-processing
    def predict_labels(self, logp, downsampling, target_steps):
        """(bs, C, T, E) log-probabilities -> int64 labels (bs, target_steps, E) (upsample, match_shape, argmax)."""
        assert logp.dim() == 4 and logp.dtype == torch.float32
        logp = logp.contiguous()
        bs, Cn, T, E = logp.shape
        labels = torch.empty(bs, target_steps, E, dtype=torch.int64, device=logp.device)
        self._check(self.lib.twog_predict_labels(logp.data_ptr(), bs, Cn, T, E, int(downsampling), int(target_steps),
                                                 labels.data_ptr(), self._stream()), 'twog_predict_labels')
        return labels

    def f1_at_k(self, y_true, y_pred, num_classes, overlap, ignore_value=None):
        """Per-sequence F1@k and validity flags for int64 (n_seq, n_steps) label matrices."""
        y_true, y_pred = y_true.to(torch.int64).contiguous(), y_pred.to(torch.int64).contiguous()
        n_seq, n_steps = y_true.shape
        dev = y_true.device
        f1 = torch.zeros(n_seq, dtype=torch.float32, device=dev)
        valid = torch.zeros(n_seq, dtype=torch.float32, device=dev)
        scratch = torch.empty(max(n_seq * n_steps, 1), dtype=torch.uint8, device=dev)
        self._check(self.lib.twog_f1_at_k(y_true.data_ptr(), y_pred.data_ptr(), n_seq, n_steps, int(num_classes),
                                          float(overlap), int(ignore_value) if ignore_value is not None else 0,
                                          int(ignore_value is not None), scratch.data_ptr(), f1.data_ptr(),
                                          valid.data_ptr(), self._stream()), 'twog_f1_at_k')
        return f1, valid


_backend = None


def get_kernels():
    """The process-wide kernel backend. Always the HIP library; raises if it is not built."""
    global _backend
    if _backend is None:
        _backend = HipKernels()
    return _backend


def _set_backend_for_tests(backend):
    """Test hook: tests/ inject a torch implementation of this interface to exercise the host logic without a GPU.
    Never called from the product path."""
    global _backend
    _backend = backend
