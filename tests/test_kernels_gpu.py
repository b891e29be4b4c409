"""GPU: every HIP kernel entry point (through the C ABI, via kernels.HipKernels) against its executable specification
(tests/fake_kernels.py, plain torch fp32 on CPU) on seeded random inputs, including strided views, ragged sizes,
masks and the split-K / grouped / batched GEMM forms. Tolerances are fp32 summation-order tolerances."""
import json
import math
import time
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import kernels as twog_kernels
from tests.fake_kernels import FakeKernels

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def K():
    twog_kernels._set_backend_for_tests(None)
    k = twog_kernels.get_kernels()
    assert k.name == 'hip'
    return k


F = FakeKernels()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def close(a, b, rtol=2e-5, atol=1e-5, what=''):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    tol = atol + rtol * b.abs().max().item() if b.numel() else atol
    assert err <= tol, f'{what}: max err {err:.3e} > tol {tol:.3e} (ref max {b.abs().max().item():.3e})'


def both(fn, bases):
    """run fn(K-like, {name: tensor}) on cpu(fake) and gpu(hip) from the same base tensors; returns (cpu, gpu) bases."""
    cpu = {k: v.clone() for k, v in bases.items()}
    gpu = {k: v.clone().to(DEV) for k, v in bases.items()}
    return cpu, gpu


# ---------------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('akm,bkm', [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize('M,N,K_', [(300, 200, 100), (70, 13, 64), (1000, 520, 264), (33, 65, 36)])
def test_gemm_layouts(K, akm, bkm, M, N, K_):
    A = rnd(K_, M) if akm else rnd(M, K_)
    B = rnd(K_, N, seed=1) if bkm else rnd(N, K_, seed=1)
    bias = rnd(N, seed=2)
    for act, acc in ((0, False), (1, True)):
        C0 = rnd(M, N, seed=3)
        c_cpu = C0.clone()
        F.gemm([dict(A=A, B=B, C=c_cpu, bias=bias, act=act, accumulate=acc)], a_kmajor=akm, b_kmajor=bkm)
        c_gpu = C0.clone().to(DEV)
        K.gemm([dict(A=A.to(DEV), B=B.to(DEV), C=c_gpu, bias=bias.to(DEV), act=act, accumulate=acc)], a_kmajor=akm,
               b_kmajor=bkm)
        close(c_gpu, c_cpu, rtol=3e-5, atol=3e-5, what=f'gemm {akm}{bkm} {M}x{N}x{K_} act{act}')


def test_gemm_unaligned_k(K):
    # K not a multiple of 4 and odd leading dimensions -> scalar load path
    M, N, K_ = 50, 40, 13
    A, B = rnd(M, K_), rnd(N, K_, seed=1)
    c_cpu = torch.zeros(M, N)
    F.gemm([dict(A=A, B=B, C=c_cpu)])
    c_gpu = torch.zeros(M, N, device=DEV)
    K.gemm([dict(A=A.to(DEV), B=B.to(DEV), C=c_gpu)])
    close(c_gpu, c_cpu, what='gemm unaligned')


def test_gemm_big_tiles_and_views(K):
    # row-strided views (column slices of wider buffers, 3-D (outer, inner, cols) rows) on the 64x64 class (25 tiles of
    # 128 do not fill the chip); the 128x128 class has its own tests below (test_gemm_128_class*)
    bs, T, E, w = 4, 30, 5, 96
    base = rnd(bs, T, E, w)
    W = rnd(640, 32, seed=1)
    out = torch.zeros(bs, T, E, 700)
    bias = rnd(640, seed=2)

    def run(Kx, base, W, out, bias):
        A = base.view(-1, w)[:, 16:48]  # (rows, 32) stride w
        C = out.view(-1, 700)[:, 20:660]
        Kx.gemm([dict(A=A, B=W, C=C, bias=bias, act=1)])
        # 3-D rows: (b, e) rows at fixed t
        A3 = base[:, 7, :, 0:32]
        C3 = out[:, 9, :, 660:692]
        Kx.gemm([dict(A=A3, B=W[:32], C=C3)])

    o_cpu = out.clone()
    run(F, base, W, o_cpu, bias)
    o_gpu = out.clone().to(DEV)
    run(K, base.to(DEV), W.to(DEV), o_gpu, bias.to(DEV))
    close(o_gpu, o_cpu, what='gemm views')


def test_gemm_grouped_batched_splitk(K):
    # grouped: several problems in one call; batched: GCN projection form; split-K: tall reduction
    A1, B1 = rnd(200, 64), rnd(96, 64, seed=1)
    A2, B2 = rnd(77, 64, seed=2), rnd(130, 64, seed=3)
    c1, c2 = torch.zeros(200, 96), torch.zeros(77, 130)
    F.gemm([dict(A=A1, B=B1, C=c1), dict(A=A2, B=B2, C=c2, act=1)])
    g1, g2 = torch.zeros(200, 96, device=DEV), torch.zeros(77, 130, device=DEV)
    K.gemm([dict(A=A1.to(DEV), B=B1.to(DEV), C=g1), dict(A=A2.to(DEV), B=B2.to(DEV), C=g2, act=1)])
    close(g1, c1, what='grouped 1')
    close(g2, c2, what='grouped 2')
    # batched GCN projection: out[b] (128, N*T) = W^T (k-major) x Z[b] rows (n,t) 2-level
    bs, T, N = 3, 7, 19
    Z, W = rnd(bs * T * N, 64), rnd(64, 128, seed=5)

    def proj(Kx, Z, W, out):
        Zv = Z.view(bs, T, N, 64).permute(0, 2, 1, 3)
        Kx.gemm([dict(A=W, B=Zv[0], C=out[0].view(128, N * T), batch=(bs, 0, T * N * 64, 128 * N * T))], a_kmajor=True)

    o_cpu = torch.zeros(bs, 128, N, T)
    proj(F, Z, W, o_cpu)
    o_gpu = torch.zeros(bs, 128, N, T, device=DEV)
    proj(K, Z.to(DEV), W.to(DEV), o_gpu)
    close(o_gpu, o_cpu, what='batched projection')
    ref = (Z.view(bs, T, N, 64) @ W).permute(0, 3, 2, 1)
    close(o_gpu, ref, what='batched projection vs einsum')
    # split-K weight gradient: dW (96, 64) = dY^T X with 30000 rows
    dY, X = rnd(30000, 96, scale=0.1), rnd(30000, 64, seed=7)
    w_cpu = torch.zeros(96, 64)
    F.gemm([dict(A=dY, B=X, C=w_cpu)], a_kmajor=True, b_kmajor=True)
    w_gpu = torch.zeros(96, 64, device=DEV)
    K.gemm([dict(A=dY.to(DEV), B=X.to(DEV), C=w_gpu)], a_kmajor=True, b_kmajor=True)
    close(w_gpu, w_cpu, rtol=1e-4, atol=1e-3, what='split-K dW')
    w_gpu2 = torch.zeros(96, 64, device=DEV)
    K.gemm([dict(A=dY.to(DEV), B=X.to(DEV), C=w_gpu2)], a_kmajor=True, b_kmajor=True)
    assert torch.equal(w_gpu, w_gpu2), 'split-K reduction must be deterministic'


# ------------------------------------------------------------------------------------------- GEMM, 128x128 tile class
# The kernels that carry the bench's roofline entry: gemm_kernel<128,128,...> in every operand layout the step uses
# (forward NN, dX NT, dW TT with split-K, GCN-style TN), the grouped-row k-major (KG, 4-wave) variants of the recurrent
# weight gradients, bias / ReLU / accumulate epilogues and splitk_reduce<128,128>. Every case ASSERTS the variant that
# ran (twog_gemm_last_class), so a change of the tile policy cannot silently move these shapes to the 64x64 class.
# Checked against an fp64 matmul (torch on the device is the checker here, never the product).
def _gemm128_run(K, akm, bkm, M, N, K_, *, bias, act, acc, ws=True, group=None, seed=0, expect=None, rtol=3e-5):
    """group=(outer, inner, skip): k-major operands are 3-D views [outer, inner, cols] cut out of [outer, inner + skip,
    wider] buffers (the 'all but the first time step of every clip' form of dW_hh, ops.py)."""
    dev = DEV
    g = torch.Generator().manual_seed(1000 + seed)

    def mk(rows, cols, kmajor, scale):
        if group is not None and kmajor:
            outer, inner, skip = group
            assert outer * inner == rows
            base = (torch.randn(outer, inner + skip, cols + 8, generator=g) * scale).to(dev)
            return base[:, skip:, 4:4 + cols]
        return (torch.randn(rows, cols, generator=g) * scale).to(dev)

    A = mk(K_, M, True, 1.0) if akm else mk(M, K_, False, 1.0)
    B = mk(K_, N, True, 0.1) if bkm else mk(N, K_, False, 0.1)
    b = (torch.randn(N, generator=g)).to(dev) if bias else None
    C0 = torch.randn(M, N, generator=g).to(dev)
    Cg = C0.clone()
    K.gemm([dict(A=A, B=B, C=Cg, bias=b, act=act, accumulate=acc)], a_kmajor=akm, b_kmajor=bkm, split_k_workspace=ws)
    cls = K.gemm_last_class()
    if expect is not None:
        want, mask = expect
        assert cls & mask == want, f'kernel class {cls:#x}, expected {want:#x} under mask {mask:#x}'
    Am = A.reshape(-1, A.shape[-1]).double()
    Bm = B.reshape(-1, B.shape[-1]).double()
    ref = (Am.t() if akm else Am) @ (Bm if bkm else Bm.t())
    if bias:
        ref = ref + b.double()
    if acc:
        ref = ref + C0.double()
    if act:
        ref = torch.relu(ref)
    err = (Cg.double() - ref).abs().max().item()
    tol = rtol * ref.abs().max().item()
    assert err <= tol, f'gemm128 akm={akm} bkm={bkm} {M}x{N}x{K_}: err {err:.3e} > {tol:.3e}'
    return cls


_LA_CHILD = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import twog_gcn_amd
from twog_gcn_amd import kernels
K = kernels.get_kernels()
out = {}
for name, (akm, bkm, M, N, Kd, bias, act, acc, seed) in dict(
        tt=(True, True, 512, 2048, 61440, False, 0, False, 4), tt2=(True, True, 1536, 2560, 30720, True, 1, True, 5),
        nt=(False, True, 20000, 1000, 1536, False, 0, False, 3), small=(True, True, 192, 192, 8192, True, 0, True, 9)).items():
    g = torch.Generator().manual_seed(1000 + seed)
    A = (torch.randn((Kd, M) if akm else (M, Kd), generator=g)).cuda()
    B = (torch.randn((Kd, N) if bkm else (N, Kd), generator=g) * 0.1).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    C = torch.randn(M, N, generator=g).cuda()
    for rep in range(2):   # twice: the tickets must be back at zero after the first launch
        Cg = C.clone()
        K.gemm([dict(A=A, B=B, C=Cg, bias=b, act=act, accumulate=acc)], a_kmajor=akm, b_kmajor=bkm)
        out[f'{name}{rep}'] = (Cg.cpu(), K.gemm_last_class())
torch.cuda.synchronize()
torch.save(out, sys.argv[2])
"""


def test_gemm_split_k_in_launch_combine_equals_the_reduce_launch_bit_for_bit(K, tmp_path):
    """TWOG_GEMM_LA=1 (the k-slices of a tile combined inside the launch by the slice that arrives last; off by default:
    measured slower, profiles/r05_splitk_in_launch_combine_ab.txt) against the default slabs + ordered-reduce launch: both add
    the slices in slice order, so every output word must be equal -- tall dW reductions (8-wave X3 class, XCD-dealt splits),
    a dX shape, a small-output shape of the 64x64 class; every case twice (the tickets return to zero)."""
    res = {}
    for la in ('0', '1'):
        f = tmp_path / f'la{la}.pt'
        env = dict(os.environ, TWOG_GEMM_LA=la)
        r = subprocess.run([sys.executable, '-c', _LA_CHILD, ROOT, str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[la] = torch.load(f)
    split_seen = False
    for k in res['0']:
        (a, ca), (b, cb) = res['0'][k], res['1'][k]
        assert ca == cb, (k, hex(ca), hex(cb))
        split_seen = split_seen or bool(ca & K.GEMM_SPLITK)
        assert torch.isfinite(b).all() and torch.equal(a, b), f'{k}: in-launch combine differs from the reduce launch'
    assert split_seen, 'no case took the split-K path'


_KU_CHILD = """
import sys, torch
sys.path.insert(0, sys.argv[1])
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
out = {}
for (M, N, Kd, seed) in ((1280, 1536, 1024, 1), (512, 1536, 512, 2), (2000, 1000, 96, 3), (128, 128, 32, 4), (640, 384, 112, 5), (256, 256, 16, 6), (384, 512, 208, 7)):
    g = torch.Generator().manual_seed(500 + seed)
    A, B, bias = torch.randn(M, Kd, generator=g).cuda(), torch.randn(N, Kd, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    C = torch.empty(M, N, device='cuda')
    K.gemm([dict(A=A, B=B, C=C, bias=bias, act=1)], split_k_workspace=False)
    ref = torch.relu(A.double() @ B.double().t() + bias.double())
    out[(M, N, Kd)] = (C.cpu(), K.gemm_last_class(), float((C.double() - ref).abs().max() / ref.abs().max()))
torch.save(out, sys.argv[2])
"""


def test_gemm_x3_128_class_two_k_tiles_per_barrier_option_is_bit_identical(K, tmp_path):
    """TWOG_X3_KU128=1 (round 6, off by default: measured slower, profiles/r06_gemm128_ku2_ab.txt): forward-form launches of the
    bf16x3 128x128 class with at most one tile per CU run gemm_x3_nn_ku2_kernel -- two k-tiles per barrier interval, the same MFMA
    sequence into the same accumulators. Every output word must equal the default kernel's, with bias + ReLU epilogue, a ragged
    problem, a K that is not a multiple of 32 (falls back to the default kernel) and a single k-pair."""
    res = {}
    for ku in ('0', '1'):
        f = tmp_path / f'ku{ku}.pt'
        r = subprocess.run([sys.executable, '-c', _KU_CHILD, ROOT, str(f)], env=dict(os.environ, TWOG_X3_KU128=ku, TWOG_X3_K2='0', TWOG_X3_PIPE='0', TWOG_GEMM_TILE='128'),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[ku] = torch.load(f)
    for k in res['0']:
        (a, ca, ea), (b, cb, eb) = res['0'][k], res['1'][k]
        assert ca & K.GEMM_TILE128 and ca & K.GEMM_X3, (k, hex(ca))
        assert ea < 2e-6 and eb < 2e-6, (k, ea, eb)
        assert torch.equal(a, b), f'{k}: two k-tiles per barrier interval changed the result'


def test_gemm_x3_128_class_sixteen_wave_tile_for_launches_of_one_tile_per_cu(K, tmp_path):
    """Round 6, TWOG_X3_K2=1 (off by default: measured no faster, profiles/r06_gemm128_two_k_groups.txt): forward-form launches
    of the bf16x3 128x128 class with at most one tile per CU (the segment level's per-step projection: 240 tiles) run
    gemm_x3_nn_k2_kernel -- 16 waves, the reduction halved between two k-groups of eight, the two partial tiles added in fixed
    order. Same products, one more fp32 addition per element: against
    fp64 both stay inside the class's 2e-6, they differ from each other by rounding only, and two runs agree bit for bit. A K
    that is not a whole pair of k-tiles, or shorter than eight k-tiles, keeps the 8-wave kernel (identical words)."""
    res = {}
    for tag, k2 in (('off', '0'), ('on', '1'), ('again', '1')):
        f = tmp_path / f'k2{tag}.pt'
        r = subprocess.run([sys.executable, '-c', _KU_CHILD, ROOT, str(f)], env=dict(os.environ, TWOG_X3_K2=k2, TWOG_X3_PIPE='0', TWOG_GEMM_TILE='128'),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(f)
    for k in res['off']:
        (a, ca, ea), (b, cb, eb), (c, _, _) = res['off'][k], res['on'][k], res['again'][k]
        assert ca & K.GEMM_TILE128 and ca & K.GEMM_X3 and cb == ca, (k, hex(ca), hex(cb))
        assert ea < 2e-6 and eb < 2e-6, (k, ea, eb)
        assert torch.equal(b, c), f'{k}: two runs of the 16-wave tile differ'
        if k[2] % 32 or k[2] < 128 or k[0] * k[1] > 256 * 128 * 128:
            assert torch.equal(a, b), k
        else:
            assert not torch.equal(a, b), f'{k}: the 16-wave kernel did not run'
            assert ((a - b).abs().max() / a.abs().max()).item() < 3e-6, k


def test_gemm_x3_128_class_fragment_reads_one_k_tile_ahead_is_bit_identical(K, tmp_path):
    """Round 6: gemm_x3_pipe_kernel reads the fragments of k-tile t + 1 before it multiplies k-tile t (two fragment sets, three
    LDS stages, one workgroup per CU) -- the same MFMA sequence into the same accumulators as gemm_x3_kernel. Every output word
    must equal the unpipelined kernel's (TWOG_X3_PIPE=0), with bias + ReLU epilogue, a ragged problem, reductions of 6, 2 and 64
    k-tiles (main loop + every remainder length)."""
    res = {}
    for pipe in ('0', '3'):
        f = tmp_path / f'pipe{pipe}.pt'
        r = subprocess.run([sys.executable, '-c', _KU_CHILD, ROOT, str(f)], env=dict(os.environ, TWOG_X3_PIPE=pipe, TWOG_X3_K2='0', TWOG_GEMM_TILE='128'),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[pipe] = torch.load(f)
    for k in res['0']:
        (a, ca, ea), (b, cb, eb) = res['0'][k], res['3'][k]
        assert ca & K.GEMM_TILE128 and ca & K.GEMM_X3, (k, hex(ca))
        assert ea < 2e-6 and eb < 2e-6, (k, ea, eb)
        assert torch.equal(a, b), f'{k}: reading the fragments one k-tile ahead changed the result'


def test_gemm_column_sums_of_a_from_the_same_pass(K):
    """twog_gemm_t::a_colsum: the dW = dY^T X launch of the bf16x3 128x128 class (k-major A and B) also returns the column sums
    of dY -- the layer's bias gradient -- from the values it stages anyway (workgroups of the first column panel add them per
    k-slice; splitk_reduce_kernel adds the slices in order). Against fp64, with and without split-K, write and accumulate, a
    ragged last row panel; C must be the bits of the launch without the request; two launches agree bit for bit; a request the
    launch would not serve is refused (rc -5), never dropped."""
    for (M, N, Kd, acc, cacc, seed, expect_split) in ((512, 2048, 61440, False, False, 1, True), (1536, 512, 30720, True, True, 2, True),
                                                      (520, 640, 8192, False, True, 3, True), (2048, 2048, 512, False, False, 4, False)):
        g = torch.Generator().manual_seed(300 + seed)
        A = (torch.randn(Kd, M, generator=g) + 0.25).to(DEV)   # a non-zero mean: the sums grow with K
        B = (torch.randn(Kd, N, generator=g) * 0.1).to(DEV)
        C0 = torch.randn(M, N, generator=g).to(DEV)
        cs0 = torch.randn(M, generator=g).to(DEV)
        p = dict(A=A, B=B, accumulate=acc)
        assert K.gemm_colsum_ok(dict(p, C=C0)), (M, N, Kd)
        Cref = C0.clone()
        K.gemm([dict(p, C=Cref)], a_kmajor=True, b_kmajor=True)
        outs = []
        for _ in range(2):
            C, cs = C0.clone(), cs0.clone()
            K.gemm([dict(p, C=C, colsum=cs, colsum_accumulate=cacc)], a_kmajor=True, b_kmajor=True)
            cls = K.gemm_last_class()
            assert cls & K.GEMM_TILE128 and cls & K.GEMM_X3 and bool(cls & K.GEMM_SPLITK) == expect_split, hex(cls)
            outs.append((C, cs))
        assert torch.equal(outs[0][0], Cref), 'the request changed C'
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        ref = A.double().sum(0) + (cs0.double() if cacc else 0.0)
        err = (outs[0][1].double() - ref).abs().max().item()
        assert err <= 3e-6 * ref.abs().max().item(), (M, N, Kd, err / ref.abs().max().item())
    # two problems in one launch, one with a request; then a shape of the 64x64 class: refused
    g = torch.Generator().manual_seed(9)
    A1, A2 = torch.randn(16384, 256, generator=g).to(DEV), torch.randn(16384, 384, generator=g).to(DEV)
    B1 = torch.randn(16384, 512, generator=g).to(DEV)
    C1, C2, cs = torch.empty(256, 512, device=DEV), torch.empty(384, 512, device=DEV), torch.zeros(384, device=DEV)
    K.gemm([dict(A=A1, B=B1, C=C1), dict(A=A2, B=B1, C=C2, colsum=cs)], a_kmajor=True, b_kmajor=True)
    ref = A2.double().sum(0)
    assert (cs.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    close(C2, (A2.double().t() @ B1.double()).float(), rtol=3e-5, atol=3e-5 * float(ref.abs().max()), what='C of the grouped launch')
    small = dict(A=torch.randn(512, 64, generator=g).to(DEV), B=torch.randn(512, 64, generator=g).to(DEV), C=torch.empty(64, 64, device=DEV))
    assert not K.gemm_colsum_ok(small)
    with pytest.raises(RuntimeError):
        K.gemm([dict(small, colsum=torch.zeros(64, device=DEV))], a_kmajor=True, b_kmajor=True)


def _gemm128_cases(K, w8=True):
    T128, W8, KG, SK = K.GEMM_TILE128, K.GEMM_WAVES8, K.GEMM_KG, K.GEMM_SPLITK
    full = T128 | W8 | KG | SK
    nosk = T128 | W8 | KG   # long reductions with a workspace may or may not be split (policy: whole rounds of the chip)
    w = W8 if w8 else 0
    # grouped k-major rows: 8-wave X3 kernels by default, the 4-wave fp32 kernels with TWOG_GEMM_X3=0 (or TWOG_GEMM_W8=0)
    kgw = w if os.environ.get('TWOG_GEMM_X3', '1') != '0' else 0
    # forward projection (NN), bias + ReLU: 480 x 4 tiles (+ ragged edge rows / columns in the second case)
    _gemm128_run(K, False, False, 61440, 512, 2048, bias=True, act=1, acc=False, expect=(T128 | w, nosk))
    _gemm128_run(K, False, False, 15361, 1500, 512, bias=True, act=0, acc=True, seed=1, expect=(T128 | w, full))
    # dX (NT), accumulate into the entity-row gradient
    _gemm128_run(K, False, True, 61440, 2048, 512, bias=False, act=0, acc=True, seed=2, expect=(T128 | w, full))
    _gemm128_run(K, False, True, 20000, 1000, 1536, bias=False, act=0, acc=False, seed=3, expect=(T128 | w | SK, full))
    # dW (TT) with deterministic split-K over 61 440 rows (+ accumulate: the in-place gradient sink route)
    _gemm128_run(K, True, True, 512, 2048, 61440, bias=False, act=0, acc=False, seed=4, expect=(T128 | w | SK, full),
                 rtol=6e-5)
    _gemm128_run(K, True, True, 1536, 2560, 30720, bias=True, act=1, acc=True, seed=5, expect=(T128 | w | SK, full),
                 rtol=6e-5)
    # TT without a workspace: no split-K, 16 x 16 tiles
    _gemm128_run(K, True, True, 2048, 2048, 1024, bias=False, act=0, acc=False, ws=False, seed=6,
                 expect=(T128 | w, full))
    # TN (k-major A, row-major B: the GCN projection form)
    _gemm128_run(K, True, False, 2048, 2100, 512, bias=True, act=0, acc=False, seed=7, expect=(T128 | w, full))
    # grouped-row k-major operands (KG): dW_hh form  1536 x 512 x (bs (T-1) E)  with split-K,
    # and a wide one without split-K
    _gemm128_run(K, True, True, 1536, 512, 8 * 59 * 8, bias=False, act=0, acc=False, group=(8, 59 * 8, 8), seed=8,
                 expect=(T128 | KG | SK | kgw, full), rtol=6e-5)
    _gemm128_run(K, True, True, 2048, 2048, 6 * 100, bias=False, act=0, acc=True, ws=False, group=(6, 100, 3), seed=9,
                 expect=(T128 | KG | (0 if (6 * 100) % 16 else kgw), full))
    # NT with a grouped k-major B only (carry form)
    _gemm128_run(K, False, True, 2048, 2048, 4 * 128, bias=False, act=0, acc=False, ws=False, group=(4, 128, 2),
                 seed=10, expect=(T128 | KG | kgw, full))


def test_gemm_128_class(K):
    _gemm128_cases(K, w8=True)


def test_gemm_128_class_determinism(K):
    # split-K slabs are summed in fixed order: two runs are bit-identical
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(30720, 1536, generator=g).to(DEV), (torch.randn(30720, 512, generator=g) * 0.1).to(DEV)
    outs = []
    for _ in range(2):
        C = torch.empty(1536, 512, device=DEV)
        K.gemm([dict(A=A, B=B, C=C)], a_kmajor=True, b_kmajor=True)
        assert K.gemm_last_class() & (K.GEMM_TILE128 | K.GEMM_SPLITK) == (K.GEMM_TILE128 | K.GEMM_SPLITK)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])


def test_gemm_128_class_four_wave_tiles():
    """The 4-wave 128x128 kernels without grouped rows are only reachable with TWOG_GEMM_W8=0 (read once per process):
    run the same cases in a child process with that switch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import tests.test_kernels_gpu as t; from twog_gcn_amd import kernels as k; '
            't._gemm128_cases(k.get_kernels(), w8=False); print("four-wave OK")')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, TWOG_GEMM_W8='0'),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'four-wave OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_gemm_128_class_split_k_dealt_by_tile_chunks():
    """Split-K launches of the 128x128 class deal whole k-splits to XCDs by default (a multiple of 8 splits, remapped in
    the kernel from the workgroup's linear index); TWOG_GEMM_XCD_SPLIT=0 (read once per process) keeps the tile-chunk deal
    with the round-filling split count. The same cases (incl. the 61 440-row reductions and the grouped-row variants) in a
    child process with that switch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import tests.test_kernels_gpu as t; from twog_gcn_amd import kernels as k; '
            't._gemm128_cases(k.get_kernels(), w8=True); print("tile-chunk deal OK")')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, TWOG_GEMM_XCD_SPLIT='0'),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'tile-chunk deal OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_gemm_64_class_is_reported(K):
    A, B, C = rnd(300, 64).to(DEV), rnd(200, 64, seed=1).to(DEV), torch.empty(300, 200, device=DEV)
    K.gemm([dict(A=A, B=B, C=C)])
    assert K.gemm_last_class() & K.GEMM_TILE128 == 0


# ----------------------------------------------------------------------------------------------------------------- GCN
@pytest.mark.parametrize('N', [1, 16, 19, 34, 50, 64])   # 50 / 64: the backward kernel's reduced-LDS path (no M copy)
def test_gcn_kernels(K, N):
    bs, T, H = 3, 6, 2
    xh = rnd(bs, T, H, 2048 + 4 * N)
    gamma, beta = rnd(4 * N, seed=1).abs() + 0.5, rnd(4 * N, seed=2)
    for training in (True, False):
        rm, rv = rnd(4 * N, seed=3) * 0.1, rnd(4 * N, seed=4).abs() + 0.5
        nbt = torch.tensor(5, dtype=torch.int64)
        rm_g, rv_g, nbt_g = rm.clone().to(DEV), rv.clone().to(DEV), nbt.clone().to(DEV)
        ab_c, mi_c = F.bn_fold(xh, N, gamma, beta, rm, rv, nbt, training)
        ab_g, mi_g = K.bn_fold(xh.to(DEV), N, gamma.to(DEV), beta.to(DEV), rm_g, rv_g, nbt_g, training)
        close(ab_g, ab_c, what=f'bn ab train={training}')
        close(mi_g, mi_c, what='bn mean/invstd')
        close(rm_g, rm, what='running_mean')
        close(rv_g, rv, what='running_var')
        assert int(nbt_g) == int(nbt)
    w1, b1 = rnd(64, 4, seed=5), rnd(64, seed=6)
    e1_c = F.gcn_embed1_fwd(xh, N, ab_c, w1, b1)
    e1_g = K.gcn_embed1_fwd(xh.to(DEV), N, ab_c.to(DEV), w1.to(DEV), b1.to(DEV))
    close(e1_g, e1_c, what='embed1 fwd')
    de1 = rnd(bs * T * N, 64, seed=7) * (e1_c > 0)
    r_c = F.gcn_embed1_bwd(xh, N, ab_c, mi_c, w1, de1)
    r_g = K.gcn_embed1_bwd(xh.to(DEV), N, ab_c.to(DEV), mi_c.to(DEV), w1.to(DEV), de1.to(DEV))
    for a, b, nm in zip(r_g, r_c, ('dw1', 'db1', 'dgamma', 'dbeta')):
        close(a, b, rtol=1e-4, atol=1e-4, what='embed1 bwd ' + nm)
    nF = bs * T
    qk, x = rnd(nF * N, 256, seed=8, scale=0.3), rnd(nF * N, 64, seed=9)
    s_c, z_c = F.gcn_attn_fwd(qk, x, nF, N)
    s_g, z_g = K.gcn_attn_fwd(qk.to(DEV), x.to(DEV), nF, N)
    close(s_g, s_c, rtol=1e-4, atol=1e-6, what='gcn attn S')
    close(z_g, z_c, rtol=1e-4, atol=1e-5, what='gcn attn Z')
    dz = rnd(nF * N, 64, seed=10)
    dx_c, dqk_c = F.gcn_attn_bwd(qk, x, s_c, dz, nF, N)
    dx_g, dqk_g = K.gcn_attn_bwd(qk.to(DEV), x.to(DEV), s_c.to(DEV), dz.to(DEV), nF, N)
    close(dx_g, dx_c, rtol=1e-4, atol=1e-5, what='gcn attn dX')
    close(dqk_g, dqk_c, rtol=1e-4, atol=1e-5, what='gcn attn dQK')
    # folded-projection MFMA kernels
    md = torch.cat([rnd(64, 64, seed=11, scale=0.05), rnd(1, 64, seed=12, scale=0.3)], 0)
    s2_c, z2_c = F.gcn_attn2_fwd(x, md, nF, N)
    s2_g, z2_g = K.gcn_attn2_fwd(x.to(DEV), md.to(DEV), nF, N)
    close(s2_g, s2_c, rtol=1e-4, atol=1e-6, what='gcn attn2 S')
    close(z2_g, z2_c, rtol=1e-4, atol=1e-5, what='gcn attn2 Z')
    dx2_c, dmd_c = F.gcn_attn2_bwd(x, md, s2_c, dz, nF, N)
    dx2_g, dmd_g = K.gcn_attn2_bwd(x.to(DEV), md.to(DEV), s2_c.to(DEV), dz.to(DEV), nF, N)
    close(dx2_g, dx2_c, rtol=1e-4, atol=1e-5, what='gcn attn2 dX')
    close(dmd_g, dmd_c, rtol=2e-4, atol=1e-4 * float(dmd_c.abs().max()), what='gcn attn2 dM|dd')
    # the fused forward kernel (geo_fused.hip): embed -> X -> folded similarity -> softmax -> aggregation, groups of frames
    w2, b2 = rnd(64, 64, seed=13, scale=0.2), rnd(64, seed=14, scale=0.2)
    for nf_sub in (nF, 1, 5):   # whole groups, a single frame, a ragged last group
        xs = xh[:, :T].reshape(bs * T, H, -1)[:nf_sub].reshape(1, nf_sub, H, -1).contiguous()
        X_c, adj_c, Z_c = F.gcn_fused_fwd(xs, N, ab_c, w1, b1, w2, b2, md)
        X_g, adj_g, Z_g = K.gcn_fused_fwd(xs.to(DEV), N, ab_c.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), md.to(DEV))
        close(X_g, X_c, rtol=2e-5, atol=1e-5, what=f'fused X ({nf_sub} frames)')
        close(adj_g, adj_c, rtol=1e-4, atol=1e-6, what='fused adjacency')
        close(Z_g, Z_c, rtol=1e-4, atol=1e-5, what='fused Z')
        Xn, adj_n, Z_n = K.gcn_fused_fwd(xs.to(DEV), N, ab_c.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), md.to(DEV), save_x=False)
        assert Xn is None and torch.equal(adj_n, adj_g) and torch.equal(Z_n, Z_g)
    # the e1 the backward pass recomputes has exactly the signs / values the fused kernel multiplied: X from the stand-alone
    # embed1 kernel + a GEMM equals the fused kernel's X to summation-order rounding
    e1_g2 = K.gcn_embed1_fwd(xh.to(DEV), N, ab_c.to(DEV), w1.to(DEV), b1.to(DEV))
    Xfull, _, _ = K.gcn_fused_fwd(xh.to(DEV), N, ab_c.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), md.to(DEV))
    close(Xfull, torch.relu(e1_g2.double() @ w2.to(DEV).double().t() + b2.to(DEV).double()).float(), rtol=2e-5, atol=1e-5, what='fused X vs embed1 + GEMM')


# --------------------------------------------------------------------------------------------------------------- BiGRU
# h = 16, 72: GEMM + gate kernel per step; 32, 96 (a partial second unit tile), 512: the fused step (gemm_gru_fwd_kernel)
@pytest.mark.parametrize('fusion', ['0', '7'])   # TWOG_GRU_FWD_FUSION: GEMM + gate kernel per step / the fused step forced
@pytest.mark.parametrize('h,bs,T', [(16, 3, 5), (72, 3, 5), (32, 5, 4), (96, 30, 4), (512, 40, 3),
                                    (512, 8, 120)])   # last: BASELINE T and width
def test_bigru(K, h, bs, T, fusion, monkeypatch):
    monkeypatch.setenv('TWOG_GRU_FWD_FUSION', fusion)   # read per call by the library; part of the chain's graph key
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '0')       # this test is about the launch-per-step path and its kernel classes
    types_c, types_g = [], []
    ws = 0.2 if T < 100 else 0.2 * math.sqrt(64.0 / h)   # long chains: keep the hidden pre-activations O(1)
    for i, E in enumerate((2, 3, 1)):
        d = dict(gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=ws), b_hh_f=rnd(3 * h, seed=20 + i),
                 w_hh_r=rnd(3 * h, h, seed=30 + i, scale=ws), b_hh_r=rnd(3 * h, seed=40 + i))
        types_c.append(d)
        types_g.append({k: v.to(DEV) for k, v in d.items()})
    res_c = F.bigru_fwd(types_c, bs, T, h)
    res_g = K.bigru_fwd(types_g, bs, T, h)
    # the variant that ran (the forward chain's last launch): the fused step where forced and served, else the gate kernel's
    # GEMM partner
    cls, fused = K.gemm_last_class(), fusion == '7' and h % 32 == 0
    assert ((cls & ~K.GEMM_X3) == K.GEMM_GRUFWD) == fused, hex(cls)
    assert not fused or bool(cls & K.GEMM_X3) == (h >= 256), hex(cls)   # wide reductions multiply on the bf16 matrix cores (X3)
    for (oc, sc), (og, sg) in zip(res_c, res_g):
        close(og, oc, rtol=1e-4, atol=1e-5, what='bigru out')
        close(sg, sc, rtol=1e-4, atol=1e-5, what='bigru save')
    bt_c, bt_g = [], []
    for i, ((oc, sc), d) in enumerate(zip(res_c, types_c)):
        dout = rnd(*oc.shape, seed=50 + i)
        bt_c.append(dict(d_out=dout, save=sc, out=oc, w_hh_f=d['w_hh_f'], w_hh_r=d['w_hh_r']))
        bt_g.append({k: v.to(DEV) for k, v in bt_c[-1].items()})
    for (gc, hc), (gg, hg) in zip(F.bigru_bwd(bt_c, bs, T, h), K.bigru_bwd(bt_g, bs, T, h)):
        close(gg, gc, rtol=1e-4, atol=1e-5, what='bigru d_gi')
        close(hg, hc, rtol=1e-4, atol=1e-5, what='bigru d_gh')


@pytest.mark.parametrize('h,bs,T,Es', [(512, 64, 5, (2, 8, 1)), (512, 8, 7, (2, 4, 1)), (512, 64, 120, (2, 8, 1)),
                                       (256, 50, 4, (2, 3, 1)), (128, 3, 3, (1, 5, 1)), (512, 37, 6, (2, 9, 2)), (64, 16, 9, (2, 9, 1))])
def test_bigru_persistent_launch_matches_the_stepwise_recurrence(K, h, bs, T, Es, monkeypatch):
    """The frame-level recurrence as ONE persistent launch (csrc/gru_persist.hip: W_hh slices in registers, steps ordered
    inside the launch by agent-scope counters) against the specification and against the launch-per-step path: outputs
    and every saved gate tensor; two runs bit-identical (fixed summation order; the hand-off is either right or the
    states are garbage)."""
    types_c, types_g = [], []
    ws = 0.2 if T < 100 else 0.2 * math.sqrt(64.0 / h)
    for i, E in enumerate(Es):
        d = dict(gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=ws), b_hh_f=rnd(3 * h, seed=20 + i),
                 w_hh_r=rnd(3 * h, h, seed=30 + i, scale=ws), b_hh_r=rnd(3 * h, seed=40 + i))
        types_c.append(d)
        types_g.append({k: v.to(DEV) for k, v in d.items()})
    res_c = F.bigru_fwd(types_c, bs, T, h)
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '0')
    res_s = K.bigru_fwd(types_g, bs, T, h)
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '1')
    called = []
    real = K.lib.twog_bigru_fwd_persistent
    monkeypatch.setattr(K.lib, 'twog_bigru_fwd_persistent', lambda *a: (called.append(1), real(*a))[1])
    res_p = K.bigru_fwd(types_g, bs, T, h)
    res_q = K.bigru_fwd(types_g, bs, T, h)
    torch.cuda.synchronize()
    assert len(called) == 2, 'the persistent launch did not run'
    for (oc, sc), (os_, ss), (op, sp), (oq, sq) in zip(res_c, res_s, res_p, res_q):
        assert torch.isfinite(op).all() and torch.isfinite(sp).all()
        close(op, oc, rtol=1e-4, atol=1e-5, what='persistent bigru out vs spec')
        close(sp, sc, rtol=1e-4, atol=1e-5, what='persistent bigru save vs spec')
        close(op, os_, rtol=2e-5, atol=2e-6, what='persistent vs stepwise out')
        assert torch.equal(op, oq) and torch.equal(sp, sq), 'two runs of the persistent launch differ'
    # the default policy: the persistent launch where every wave owns at most one row tile (small batches), else per step
    monkeypatch.delenv('TWOG_BIGRU_PERSIST')
    called.clear()
    K.bigru_fwd(types_g, bs, T, h)
    assert len(called) == int(K.last_bigru_persistent), (bs, Es, called)
    if h == 512 and bs in (8, 64):
        assert K.last_bigru_persistent == (bs == 8), (bs, Es)


@pytest.mark.parametrize('h,bs,T,Es', [(512, 8, 7, (2, 4, 1)), (512, 8, 120, (2, 4, 1)), (256, 20, 4, (2, 3, 1)), (128, 3, 3, (1, 5, 1)),
                                       (512, 5, 6, (2, 9, 2)), (64, 16, 9, (2, 9, 1))])
def test_bigru_persistent_backward_matches_the_stepwise_recurrence(K, h, bs, T, Es, monkeypatch):
    """Backward through time as one persistent launch (small-batch form, csrc/gru_persist.hip) against the specification
    and the launch-per-step path: d_gi and d_gh of every type; two runs bit-identical."""
    types_c = []
    ws = 0.2 if T < 100 else 0.2 * math.sqrt(64.0 / h)
    for i, E in enumerate(Es):
        types_c.append(dict(gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=ws), b_hh_f=rnd(3 * h, seed=20 + i),
                            w_hh_r=rnd(3 * h, h, seed=30 + i, scale=ws), b_hh_r=rnd(3 * h, seed=40 + i)))
    res_c = F.bigru_fwd(types_c, bs, T, h)
    bt_c, bt_g = [], []
    for i, ((oc, sc), d) in enumerate(zip(res_c, types_c)):
        bt_c.append(dict(d_out=rnd(*oc.shape, seed=50 + i), save=sc, out=oc, w_hh_f=d['w_hh_f'], w_hh_r=d['w_hh_r']))
        bt_g.append({k: v.to(DEV) for k, v in bt_c[-1].items()})
    want = F.bigru_bwd(bt_c, bs, T, h)
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '0')
    step = K.bigru_bwd(bt_g, bs, T, h)
    assert not K.last_bigru_bwd_persistent
    monkeypatch.delenv('TWOG_BIGRU_PERSIST')
    p1 = K.bigru_bwd(bt_g, bs, T, h)
    assert K.last_bigru_bwd_persistent, 'the persistent backward did not run'
    p2 = K.bigru_bwd(bt_g, bs, T, h)
    torch.cuda.synchronize()
    for (gc, hc), (gs, hs_), (g1, h1), (g2, h2) in zip(want, step, p1, p2):
        assert torch.isfinite(g1).all() and torch.isfinite(h1).all()
        close(g1, gc, rtol=1e-4, atol=1e-5, what='persistent bigru d_gi vs spec')
        close(h1, hc, rtol=1e-4, atol=1e-5, what='persistent bigru d_gh vs spec')
        close(g1, gs, rtol=5e-5, atol=5e-6, what='persistent vs stepwise d_gi')
        assert torch.equal(g1, g2) and torch.equal(h1, h2), 'two runs of the persistent backward differ'


def test_bigru_persistent_hand_offs_hold_under_uneven_load(K):
    """The persistent launches exchange states between workgroups through write-through stores, agent-scope counters and
    sc1 loads. Hand-off mistakes hide on an idle chip (uniform timing, cold L1s): here the same forward + backward (8
    clips, T = 120, h = 512) runs while a second stream keeps part of the chip busy with GEMMs of changing sizes -- the
    persistent workgroups start staggered and progress unevenly -- and every word of every output must equal the
    solo run's, five times over."""
    bs, T, h, Es = 8, 120, 512, (2, 4, 1)
    ws = 0.2 * math.sqrt(64.0 / h)
    types = []
    for i, E in enumerate(Es):
        types.append({k: v.to(DEV) for k, v in dict(
            gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=ws), b_hh_f=rnd(3 * h, seed=20 + i),
            w_hh_r=rnd(3 * h, h, seed=30 + i, scale=ws), b_hh_r=rnd(3 * h, seed=40 + i)).items()})

    def run():
        fw = K.bigru_fwd(types, bs, T, h)
        assert K.last_bigru_persistent
        bt = [dict(d_out=rnd(bs, T, E, 2 * h, seed=50 + i).to(DEV), save=sv, out=o, w_hh_f=y['w_hh_f'], w_hh_r=y['w_hh_r'])
              for i, ((o, sv), y, E) in enumerate(zip(fw, types, Es))]
        bw = K.bigru_bwd(bt, bs, T, h)
        assert K.last_bigru_bwd_persistent
        return [t for pair in fw for t in pair] + [t for pair in bw for t in pair]

    solo = [t.clone() for t in run()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    mats = [torch.randn(n, n, device=DEV) for n in (512, 1024, 1536, 2048, 3072)]
    for rep in range(5):
        with torch.cuda.stream(side):
            for j in range(40):
                m = mats[(rep + j) % len(mats)]
                torch.mm(m, m)
        got = run()
        torch.cuda.synchronize()
        for a, b in zip(got, solo):
            assert torch.equal(a, b), f'repetition {rep}: a persistent launch under load differs from the solo run'


def _persist_rig(K):
    bs, T, h, Es = 8, 24, 512, (2, 4, 1)
    types = []
    for i, E in enumerate(Es):
        types.append({k: v.to(DEV) for k, v in dict(
            gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=0.2), b_hh_f=rnd(3 * h, seed=20 + i),
            w_hh_r=rnd(3 * h, h, seed=30 + i, scale=0.2), b_hh_r=rnd(3 * h, seed=40 + i)).items()})
    seg = _seg_params(DEV, bs, T, 2, 4, h, (True, True, True, True), True)
    seg_keys = ['hs_h', 'hs_o', 'save_h', 'save_o', 'msrc_h', 'msrc_o', 'mg_h', 'mg_o', 'att']
    dseg = (rnd(bs, T, 2, 2 * h, seed=61).to(DEV), rnd(bs, T, 4, 2 * h, seed=62).to(DEV))

    def run():
        fw = K.bigru_fwd(types, bs, T, h)
        pf = K.last_bigru_persistent
        bt = [dict(d_out=rnd(bs, T, E, 2 * h, seed=50 + i).to(DEV), save=sv, out=o, w_hh_f=y['w_hh_f'], w_hh_r=y['w_hh_r'])
              for i, ((o, sv), y, E) in enumerate(zip(fw, types, Es))]
        bw = K.bigru_bwd(bt, bs, T, h)
        pb = K.last_bigru_bwd_persistent
        sb = K.segrnn_fwd(seg)
        ps = K.last_segrnn_persistent
        so = K.segrnn_bwd(seg, sb, dseg[0], dseg[1])
        psb = K.last_segrnn_bwd_persistent
        torch.cuda.synchronize()
        return ([t for pair in fw for t in pair] + [t for pair in bw for t in pair] + [sb[k] for k in seg_keys] +
                [so[k] for k in sorted(so)]), (pf, pb, ps, psb)

    return run


def test_persistent_launches_fail_soft_and_the_pass_is_rerun_per_step(K, monkeypatch):
    """VERDICT r04 weak #2 / ADVICE r04: a persistent launch whose grid cannot make progress must neither hang nor trap.
    TWOG_PERSIST_SPIN_LIMIT=0 is the library's test hook: every inter-workgroup wait gives up at once, the launch sets its
    error word and every wave leaves. Checked at once (the first launches on a device, and TWOG_PERSIST_CHECK=sync): the host
    sees the word and re-runs the pass on the launch-per-step path before anything consumed the outputs -- results = the
    launch-per-step path's, bit for bit; the process lives; the counters say what ran; the device gets no persistent launch
    for PERSISTENT_BACKOFF calls; with the limit restored the persistent launches run again and give what they gave."""
    from twog_gcn_amd.kernels import HipKernels
    run = _persist_rig(K)
    HipKernels._backoff.clear()
    HipKernels._lazy.clear()
    monkeypatch.setenv('TWOG_PERSIST_CHECK', 'sync')
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '0')
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')
    want, ran = run()
    assert ran == (False, False, False, False)
    monkeypatch.delenv('TWOG_BIGRU_PERSIST')
    monkeypatch.delenv('TWOG_SEG_PERSIST')
    good, ran = run()
    assert ran == (True, True, True, True), 'the persistent launches did not run on an idle device'
    dev_i = torch.cuda.current_device()
    monkeypatch.setenv('TWOG_PERSIST_SPIN_LIMIT', '0')
    n0 = HipKernels.persistent_fallbacks
    got, ran = run()   # the forward BiGRU launch is tried first, gives up, the device backs off: the other two are not tried
    assert ran == (False, False, False, False)
    assert HipKernels.persistent_fallbacks == n0 + 1, 'exactly one launch was tried, gave up and was re-run'
    assert 0 < HipKernels._backoff.get(dev_i, 0) <= HipKernels.PERSISTENT_BACKOFF
    for a, b in zip(got, want):
        assert torch.isfinite(a).all() and torch.equal(a, b), 'a re-run pass differs from the launch-per-step path'
    # the backward BiGRU launch and the segment launch as the FIRST persistent launch of a call
    for env in ('bwd', 'seg', 'segb'):
        HipKernels._backoff.clear()
        n0 = HipKernels.persistent_fallbacks
        real_allowed = HipKernels.persistent_allowed
        calls = []

        def allowed(self, dev, env=env, calls=calls):   # lets only the launch under test through
            calls.append(1)
            import inspect
            caller = inspect.stack()[1].function
            want_caller = {'bwd': 'bigru_bwd', 'seg': 'segrnn_fwd', 'segb': 'segrnn_bwd'}[env]
            return caller == want_caller and real_allowed(self, dev)

        monkeypatch.setattr(HipKernels, 'persistent_allowed', allowed)
        got, ran = run()
        monkeypatch.setattr(HipKernels, 'persistent_allowed', real_allowed)
        assert ran == (False, False, False, False) and HipKernels.persistent_fallbacks == n0 + 1, (env, ran)
        for a, b in zip(got, want):
            assert torch.equal(a, b), env
    # limit restored: persistent again, same results as before
    monkeypatch.delenv('TWOG_PERSIST_SPIN_LIMIT')
    HipKernels._backoff.clear()
    again, ran = run()
    assert ran == (True, True, True, True)
    for a, b in zip(again, good):
        assert torch.equal(a, b)


def test_persistent_launch_failure_found_at_the_end_of_the_pass_raises_and_the_next_calls_recover(K, monkeypatch):
    """After PERSIST_SYNC_CALLS clean launches the error word is read at the END of the pass (no pipeline drain per launch).
    A failure found that late cannot be repaired behind the caller's back: verify_persistent raises -- process and context
    alive -- the device backs off, and the next pass runs per step with correct results."""
    from twog_gcn_amd.kernels import HipKernels
    run = _persist_rig(K)
    HipKernels._backoff.clear()
    HipKernels._lazy.clear()
    monkeypatch.setenv('TWOG_PERSIST_CHECK', 'lazy')
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '0')
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')
    want, _ = run()
    monkeypatch.delenv('TWOG_BIGRU_PERSIST')
    monkeypatch.delenv('TWOG_SEG_PERSIST')
    good, ran = run()
    assert ran == (True, True, True, True)
    K.verify_persistent()            # clean launches: nothing to report
    monkeypatch.setenv('TWOG_PERSIST_SPIN_LIMIT', '0')
    n0 = HipKernels.persistent_late_failures
    _, ran = run()
    assert ran == (True, True, True, True)   # nobody has looked yet
    with pytest.raises(RuntimeError, match='could not keep its grid resident'):
        K.verify_persistent()
    assert HipKernels.persistent_late_failures == n0 + 1
    K.verify_persistent()            # reported once
    got, ran = run()                 # backing off: per step, correct
    assert ran == (False, False, False, False)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    HipKernels._backoff.clear()
    HipKernels._lazy.clear()


def test_guard_launch_poisons_the_outputs_of_a_pass_whose_persistent_launch_gave_up(K, monkeypatch):
    """When a backward pass follows, the forward pass does not wait for its deferred error words (that wait drains the stream
    and the backward pass starts on an idle device: ~1 ms per 8-clip step): one guard launch behind them turns the pass's
    outputs into NaN if any word is set, and the words are read -- and raise -- at the end of the backward pass. Here: a
    clean pass leaves the outputs alone; with a forced give-up (TWOG_PERSIST_SPIN_LIMIT=0) every guarded tensor is NaN and
    verify_persistent raises afterwards."""
    from twog_gcn_amd.kernels import HipKernels
    run = _persist_rig(K)
    HipKernels._backoff.clear()
    HipKernels._lazy.clear()
    monkeypatch.setenv('TWOG_PERSIST_CHECK', 'lazy')
    outs = [torch.ones(1000, device=DEV), torch.full((3, 7), 2.0, device=DEV)]
    _, ran = run()
    assert ran == (True, True, True, True) and len(HipKernels._lazy[0]) == 4
    assert K.guard_persistent(torch.device(DEV), outs)
    torch.cuda.synchronize()
    assert float(outs[0].sum()) == 1000.0 and float(outs[1].sum()) == 42.0
    K.verify_persistent()
    monkeypatch.setenv('TWOG_PERSIST_SPIN_LIMIT', '0')
    _, ran = run()
    assert ran == (True, True, True, True)
    assert K.guard_persistent(torch.device(DEV), outs)
    torch.cuda.synchronize()
    assert bool(torch.isnan(outs[0]).all()) and bool(torch.isnan(outs[1]).all())
    with pytest.raises(RuntimeError, match='could not keep its grid resident'):
        K.verify_persistent()
    HipKernels._backoff.clear()
    HipKernels._lazy.clear()


def test_persistent_launch_beside_a_tenant_that_holds_compute_units(K, monkeypatch):
    """The real thing: another stream holds half of the compute units (twog_debug_occupy: 128 workgroups with 100 KB of LDS
    each, for 60 ms), so half of a persistent grid is resident and waits for the half that is not. With a spin limit of
    ~2 ms the resident waves give up, the grid drains, the host re-runs the pass per step: correct results, no hang, no trap;
    the tenant finishes undisturbed."""
    from twog_gcn_amd.kernels import HipKernels
    run = _persist_rig(K)
    HipKernels._backoff.clear()
    HipKernels._lazy.clear()
    monkeypatch.setenv('TWOG_PERSIST_CHECK', 'sync')
    monkeypatch.setenv('TWOG_BIGRU_PERSIST', '0')
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')
    want, _ = run()
    monkeypatch.delenv('TWOG_BIGRU_PERSIST')
    monkeypatch.delenv('TWOG_SEG_PERSIST')
    monkeypatch.setenv('TWOG_PERSIST_SPIN_LIMIT', '2000')
    side = torch.cuda.Stream()
    n0 = HipKernels.persistent_fallbacks
    t0 = time.time()
    with torch.cuda.stream(side):
        K.debug_occupy(128, 100 * 1024, 60000)
    got, ran = run()
    torch.cuda.synchronize()
    assert time.time() - t0 < 30.0
    assert HipKernels.persistent_fallbacks >= n0 + 1, 'no persistent launch gave up beside the tenant'
    for a, b in zip(got, want):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    HipKernels._backoff.clear()


def test_masked_side_stream_runs_the_same_kernels_on_a_subset_of_the_compute_units(K):
    """twog_stream_create_masked (the TWOG_SIDE_CUS experiment of ops.tggcn_backward: weight-gradient GEMMs on 80 of the 256
    CUs beside a launch-per-step recurrence; measured slower and off by default): a GEMM issued on the masked
    stream gives the bits of the same launch on the caller's stream, the join orders it before the caller's next launch,
    and a second request for the same mask returns the cached stream."""
    dev = torch.device(DEV)
    side = K.side_stream(dev, 80)
    if side is None:
        pytest.skip('the runtime refused hipExtStreamCreateWithCUMask')
    g = torch.Generator().manual_seed(21)
    A, B = torch.randn(4096, 512, generator=g).to(DEV), (torch.randn(2048, 512, generator=g) * 0.1).to(DEV)
    ref = torch.empty(4096, 2048, device=DEV)
    K.gemm([dict(A=A, B=B, C=ref)])
    out = torch.zeros(4096, 2048, device=DEV)
    torch.cuda.synchronize()
    with side:
        K.gemm([dict(A=A, B=B, C=out)])
    side.join()
    doubled = out * 2          # on the caller's stream, behind the join
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(doubled, ref * 2)
    assert K.side_stream(dev, 80).stream is side.stream


def test_tape_run_replays_recorded_calls_with_affine_descriptors(K):
    """twog_tape_run: two consecutive steps of a loop are recorded (descriptor arrays kept, nothing issued); step a + k is
    run with every 64-bit descriptor word a + k (b - a). A GEMM, a gate step and row operations over per-step slots of
    [steps][...] buffers: six replayed steps must equal six issued ones, and a third recorded step must equal the rule."""
    n, rows, kdim, cols = 6, 40, 64, 48
    A, B = rnd(n, rows, kdim, seed=1).to(DEV), rnd(cols, kdim, seed=2, scale=0.2).to(DEV)
    bias = rnd(cols, seed=3).to(DEV)
    src = rnd(n, rows, cols, seed=4).to(DEV)

    def step(t, C, acc, r1):
        K.gemm([dict(A=A[t], B=B, C=C[t], bias=bias, act=1)])
        K.rowops([('add', src[t], acc[t]), ('relu_bwd', src[t], C[t], r1[t])])

    def buffers():
        return (torch.zeros(n, rows, cols, device=DEV), torch.ones(n, rows, cols, device=DEV), torch.zeros(n, rows, cols, device=DEV))

    want = buffers()
    for t in range(n):
        step(t, *want)
    got = buffers()
    tapes = []
    for t in range(3):
        K.tape_begin()
        step(t, *got)
        tapes.append(K.tape_end())
    torch.cuda.synchronize()
    assert all(float(b.abs().sum()) == float(w0) for b, w0 in zip(got, (0.0, n * rows * cols, 0.0))), 'recording must not issue anything'
    assert K.tape_matches(tapes[0], tapes[1], tapes[2], 2)
    assert not K.tape_matches(tapes[0], tapes[1], tapes[1], 2)
    K.tape_run(tapes[0], tapes[1], 0, n, DEV)
    torch.cuda.synchronize()
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)


# ---------------------------------------------------------------------------------------------------- entity attention
def _attn_case(dev, H, O, D, h, n_inst, ipc, geo, recv_mask, seed=0):
    t = lambda *s, sd=0: rnd(*s, seed=seed + sd).to(dev)
    W = 3 * h
    d = dict(feat_h=t(n_inst * H, D + 8, sd=1)[:, :D], feat_o=t(n_inst * O, D, sd=2),
             msg_hh=t(n_inst * H, 2 * h, sd=3)[:, :h], msg_ho=t(n_inst * H, 2 * h, sd=4)[:, h:],
             msg_oh=t(n_inst * O, h, sd=5), msg_oo=t(n_inst * O, h, sd=6),
             out_hh=torch.zeros(n_inst * H, W, device=dev)[:, :h], out_oh=torch.zeros(n_inst * H, W, device=dev)[:, h:2 * h],
             out_ho=torch.zeros(n_inst * O, W, device=dev)[:, :h], out_oo=torch.zeros(n_inst * O, W, device=dev)[:, 2 * h:],
             att=torch.zeros(n_inst, H * H + 2 * H * O + O * O, device=dev), n_inst=n_inst, inst_per_clip=ipc, H=H, O=O,
             D=D, hidden=h, scale=1.0 / math.sqrt(D), recv_mask_ho=recv_mask)
    mask = (rnd(n_inst // ipc, O, seed=seed + 7) > -0.3).float()
    mask[0] = 0.0  # a clip with only virtual objects (NaN -> 0 path)
    d['obj_mask'] = mask.to(dev)
    if geo:
        d.update(msg_so=t(n_inst, h, sd=8), msg_sh=t(n_inst, h, sd=9),
                 out_so=torch.zeros(n_inst * O, h, device=dev), out_sh=torch.zeros(n_inst * H, h, device=dev))
    return d


@pytest.mark.parametrize('H,O,D,h,ipc,geo,rm', [(2, 4, 64, 32, 3, True, 1), (1, 5, 32, 32, 1, False, 0),
                                                (2, 9, 128, 64, 2, True, 1), (2, 8, 1024, 512, 2, True, 1)])
def test_entity_attention(K, H, O, D, h, ipc, geo, rm):
    n_inst = 6
    dc = _attn_case('cpu', H, O, D, h, n_inst, ipc, geo, rm)
    dg = _attn_case(DEV, H, O, D, h, n_inst, ipc, geo, rm)
    if H == 1:
        for d in (dc, dg):
            d['msg_hh'] = None
            d.pop('out_hh')
    F.attn_fwd([dc])
    K.attn_fwd([dg])
    close(dg['att'], dc['att'], rtol=1e-4, atol=1e-6, what='att weights')
    for k in dc:
        if k.startswith('out_'):
            close(dg[k], dc[k], rtol=1e-4, atol=1e-5, what=k)
    assert not torch.isnan(dg['att']).any()

    def bwd(dev, d, seed=100):
        t = lambda *s, sd=0: rnd(*s, seed=seed + sd).to(dev)
        b = dict(f=d, dfeat_accumulate=1, relu_mask_dmsg=1, dfeat_h=t(n_inst * H, D, sd=1), dfeat_o=t(n_inst * O, D, sd=2))
        for i, (rel, R) in enumerate((('hh', H), ('oh', H), ('ho', O), ('oo', O), ('so', O), ('sh', H))):
            if d.get('msg_' + rel) is None:
                continue
            b['dout_' + rel] = t(n_inst * R, h, sd=10 + i)
            S_ = {'hh': H, 'ho': H, 'oh': O, 'oo': O, 'so': 0, 'sh': 0}[rel]
            b['dmsg_' + rel] = torch.zeros(n_inst * S_ if S_ else n_inst, h, device=dev)
        return b

    bc, bg = bwd('cpu', dc), bwd(DEV, dg)
    F.attn_bwd([bc])
    K.attn_bwd([bg])
    for k in bc:
        if k.startswith('dmsg_') or k.startswith('dfeat_h') or k.startswith('dfeat_o'):
            close(bg[k], bc[k], rtol=2e-4, atol=2e-5, what=k)


# ---------------------------------------------------------------------------------------------- segment-level recurrence
def _seg_params(dev, bs, T, H, O, h, rels, msg_segment=True, seed=0):
    w_sc = 0.3 if h <= 64 else 0.3 * math.sqrt(64.0 / h)   # keep pre-activations O(1) at full width
    if T > 50:
        w_sc = 0.4 / math.sqrt(h)   # long chains: contractive dynamics, so rounding differences do not amplify over T
    t = lambda *s, sd=0, sc=None: rnd(*s, seed=seed + sd, scale=w_sc if sc is None else sc).to(dev)
    rel_hh, rel_ho, rel_oh, rel_oo = rels
    nmh, nmo = int(rel_hh) + int(rel_oh), int(rel_ho) + int(rel_oo)
    nsh, nso = int(rel_hh) + int(rel_ho), int(rel_oh) + int(rel_oo)
    fw_h, fw_o = 3 * h, 4 * h
    wih_h = [t(3 * h, fw_h + nmh * h, sd=1 + d) for d in range(2)]
    wih_o = [t(3 * h, fw_o + nmo * h, sd=3 + d) for d in range(2)]
    mask = torch.ones(bs, O)
    mask[0, O - 1] = 0
    if bs > 1:
        mask[1] = 0
    p = dict(bs=bs, T=T, H=H, O=O, hidden=h, msg_segment=msg_segment, rel_hh=rel_hh, rel_ho=rel_ho, rel_oh=rel_oh,
             rel_oo=rel_oo, att_scale=1 / math.sqrt(h), gi_h=t(bs, T, H, 6 * h, sd=5, sc=1.0), gi_o=t(bs, T, O, 6 * h, sd=6, sc=1.0),
             u_h=(rnd(bs, T, H, seed=seed + 7) > 0).float().to(dev), u_o=(rnd(bs, T, O, seed=seed + 8) > 0).float().to(dev),
             obj_mask=mask.to(dev),
             w_hh_h=[t(3 * h, h, sd=9 + d) for d in range(2)], b_hh_h=[t(3 * h, sd=11 + d) for d in range(2)],
             w_hh_o=[t(3 * h, h, sd=13 + d) for d in range(2)], b_hh_o=[t(3 * h, sd=15 + d) for d in range(2)],
             w_ihm_h=[w[:, fw_h:] for w in wih_h], w_ihm_o=[w[:, fw_o:] for w in wih_o],
             ld_ih_h=fw_h + nmh * h, ld_ih_o=fw_o + nmo * h,
             w_smsg_h=t(max(nsh, 1) * h, h, sd=17)[:nsh * h], b_smsg_h=t(max(nsh, 1) * h, sd=18)[:nsh * h],
             w_smsg_o=t(max(nso, 1) * h, h, sd=19)[:nso * h], b_smsg_o=t(max(nso, 1) * h, sd=20)[:nso * h])
    p['_keep'] = (wih_h, wih_o)
    return p


@pytest.mark.parametrize('bs,T,H,O,h,rels,msg', [(3, 5, 2, 4, 16, (True, True, True, True), True),
                                                 (2, 4, 1, 5, 32, (False, True, True, True), True),
                                                 (2, 3, 2, 3, 16, (True, True, True, True), False),
                                                 (4, 6, 2, 8, 64, (True, True, True, True), True),
                                                 # fused forward step with a partial second unit tile, 70 object rows
                                                 (14, 4, 2, 5, 96, (True, False, True, True), True),
                                                 # full width: 8 column tiles, several row tiles -- every partial slot of
                                                 # the fused gate-backward epilogue (with and without segment messages)
                                                 (24, 3, 2, 8, 512, (True, True, True, True), True),
                                                 (24, 3, 2, 8, 512, (True, True, True, True), False),
                                                 # BASELINE length and width: a 120-step chain, forward and BPTT
                                                 (4, 120, 2, 8, 512, (True, True, True, True), True)])
@pytest.mark.parametrize('fusion', ['0', '7'])
def test_segment_recurrence(K, bs, T, H, O, h, rels, msg, fusion, monkeypatch):
    monkeypatch.setenv('TWOG_GRU_FWD_FUSION', fusion)
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')   # this test is about the launch-per-step path and its kernel classes
    pc = _seg_params('cpu', bs, T, H, O, h, rels, msg)
    pg = _seg_params(DEV, bs, T, H, O, h, rels, msg)
    bc = F.segrnn_fwd(pc)
    bg = K.segrnn_fwd(pg)
    cls, fused = K.gemm_last_class(), fusion == '7' and h % 32 == 0
    assert ((cls & ~K.GEMM_X3) == K.GEMM_GRUFWD) == fused, hex(cls)
    assert not fused or bool(cls & K.GEMM_X3) == (h >= 256), hex(cls)   # wide reductions multiply on the bf16 matrix cores (X3)
    keys = ['hs_h', 'hs_o', 'save_h', 'save_o'] + (['msrc_h', 'msrc_o', 'mg_h', 'mg_o', 'att'] if msg else [])
    for k in keys:
        close(bg[k], bc[k], rtol=2e-4, atol=2e-5, what='segrnn fwd ' + k)
    dh_h, dh_o = rnd(bs, T, H, 2 * h, seed=31), rnd(bs, T, O, 2 * h, seed=32)
    oc = F.segrnn_bwd(pc, bc, dh_h, dh_o)
    # feed the GPU backward with the CPU forward buffers so that the comparison isolates the backward kernels
    bg2 = {k: v.to(DEV) for k, v in bc.items()}
    for k in bg:
        if k not in bg2:
            bg2[k] = bg[k]
    og = K.segrnn_bwd(pg, bg2, dh_h.to(DEV), dh_o.to(DEV))
    for k in oc:
        if k.startswith('d_pre') and not msg:
            continue  # unused scratch when message_segment is off
        close(og[k], oc[k], rtol=3e-4, atol=3e-5, what='segrnn bwd ' + k)


# the segment-level recurrence as ONE persistent launch (csrc/seg_persist.hip): BASELINE configs[1] (8 clips, MPHOI layout,
# h = 512, T = 120), configs[4] (16 clips, Bimanual layout, h = 64: eight chunks of two clips), configs[0] (one CAD-120 clip),
# a ragged last chunk, all objects of a clip masked, h = 128 / 256
SEG_PERSIST_SHAPES = [(8, 120, 2, 4, 512), (16, 120, 2, 9, 64), (1, 20, 1, 5, 512), (8, 7, 2, 4, 512), (5, 9, 2, 3, 128),
                      (3, 6, 1, 5, 256), (13, 5, 2, 9, 64), (4, 6, 2, 8, 64), (2, 1, 2, 4, 128)]


@pytest.mark.parametrize('bs,T,H,O,h', SEG_PERSIST_SHAPES)
def test_segment_recurrence_persistent_launch_matches_the_stepwise_recurrence(K, bs, T, H, O, h, monkeypatch):
    """Forward segment recurrence in one launch against the specification (the torch test double) and against the
    launch-per-step path: states, saved gate activations, sender messages, aggregated messages, attention weights; two
    runs bit-identical (fixed summation order; a wrong hand-off gives garbage, not noise)."""
    from twog_gcn_amd.kernels import HipKernels
    HipKernels._backoff.clear()
    monkeypatch.setenv('TWOG_PERSIST_CHECK', 'sync')
    rels = (True, True, True, True)
    pc = _seg_params('cpu', bs, T, H, O, h, rels, True)
    pg = _seg_params(DEV, bs, T, H, O, h, rels, True)
    bc = F.segrnn_fwd(pc)
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')
    bs_ = K.segrnn_fwd(pg)
    assert not K.last_segrnn_persistent
    monkeypatch.delenv('TWOG_SEG_PERSIST')
    b1 = K.segrnn_fwd(pg)
    assert K.last_segrnn_persistent, 'the persistent launch did not run'
    b2 = K.segrnn_fwd(pg)
    torch.cuda.synchronize()
    for k in ['hs_h', 'hs_o', 'save_h', 'save_o', 'msrc_h', 'msrc_o', 'mg_h', 'mg_o', 'att']:
        assert torch.isfinite(b1[k]).all(), k
        close(b1[k], bc[k], rtol=2e-4, atol=2e-5, what='persistent segrnn fwd vs spec: ' + k)
        close(b1[k], bs_[k], rtol=5e-5, atol=5e-6, what='persistent vs stepwise: ' + k)
        assert torch.equal(b1[k], b2[k]), 'two runs of the persistent launch differ: ' + k
    # backward through time: the persistent launch against the specification (on the specification's forward buffers, so
    # that the comparison isolates the backward kernel), against the launch-per-step path, and against itself
    dh_hc, dh_oc = rnd(bs, T, H, 2 * h, seed=31), rnd(bs, T, O, 2 * h, seed=32)
    dh_h, dh_o = dh_hc.to(DEV), dh_oc.to(DEV)
    oc = F.segrnn_bwd(pc, bc, dh_hc, dh_oc)
    bg2 = {k: v.to(DEV) for k, v in bc.items()}
    for k in b1:
        if k not in bg2:
            bg2[k] = b1[k]
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')
    o0 = K.segrnn_bwd(pg, bg2, dh_h, dh_o)
    assert not K.last_segrnn_bwd_persistent
    monkeypatch.delenv('TWOG_SEG_PERSIST')
    o1 = K.segrnn_bwd(pg, bg2, dh_h, dh_o)
    assert K.last_segrnn_bwd_persistent, 'the persistent backward launch did not run'
    o2 = K.segrnn_bwd(pg, bg2, dh_h, dh_o)
    torch.cuda.synchronize()
    for k in oc:
        assert torch.isfinite(o1[k]).all(), k
        close(o1[k], oc[k], rtol=3e-4, atol=3e-5, what='persistent segrnn bwd vs spec: ' + k)
        close(o1[k], o0[k], rtol=1e-4, atol=1e-5, what='persistent vs stepwise bwd: ' + k)
        assert torch.equal(o1[k], o2[k]), 'two runs of the persistent backward launch differ: ' + k
    # and the stepwise backward on the persistent forward's own buffers equals the one on the stepwise forward's
    monkeypatch.setenv('TWOG_SEG_PERSIST', '0')
    oa, ob = K.segrnn_bwd(pg, b1, dh_h, dh_o), K.segrnn_bwd(pg, bs_, dh_h, dh_o)
    for k in oa:
        close(oa[k], ob[k], rtol=3e-4, atol=3e-5, what='segrnn bwd on persistent forward buffers: ' + k)


def test_segment_recurrence_persistent_hand_offs_hold_under_uneven_load(K, monkeypatch):
    """Same rig as the frame-level test: the persistent segment forward (8 clips, T = 120, h = 512) while a second stream
    keeps part of the chip busy with GEMMs of changing sizes; every word of every output must equal the solo run's."""
    monkeypatch.setenv('TWOG_PERSIST_CHECK', 'sync')
    pg = _seg_params(DEV, 8, 120, 2, 4, 512, (True, True, True, True), True)
    keys = ['hs_h', 'hs_o', 'save_h', 'save_o', 'msrc_h', 'msrc_o', 'mg_h', 'mg_o', 'att']

    dh_h, dh_o = rnd(8, 120, 2, 1024, seed=31).to(DEV), rnd(8, 120, 4, 1024, seed=32).to(DEV)

    def run():
        b = K.segrnn_fwd(pg)
        assert K.last_segrnn_persistent
        o = K.segrnn_bwd(pg, b, dh_h, dh_o)
        assert K.last_segrnn_bwd_persistent
        return [b[k] for k in keys] + [o[k] for k in sorted(o)]

    solo = [t.clone() for t in run()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    mats = [torch.randn(n, n, device=DEV) for n in (512, 1024, 1536, 2048, 3072)]
    for rep in range(5):
        with torch.cuda.stream(side):
            for j in range(40):
                m = mats[(rep + j) % len(mats)]
                torch.mm(m, m)
        got = run()
        torch.cuda.synchronize()
        for k, a, b in zip(keys + ['bwd'] * 16, got, solo):
            assert torch.equal(a, b), f'repetition {rep}: {k} of the persistent launch under load differs from the solo run'


def test_persistent_hand_offs_hold_under_jitter(tmp_path):
    """the four persistent launches (BiGRU forward / backward, segment forward / backward) at the
    BASELINE configs[0] / [1] / [4] shapes, 200 times each, in the JITTER build of the library (`make jitter`:
    persist_common.h::twog_jitter, a pseudo-random pause of 0 ... ~4 us per wave in front of every publish and every poll, so
    the order in which workgroups reach their hand-offs changes from step to step) -- every word of every output equal to
    the shipped library's. A hand-off that only holds by timing shows as a wrong word; the shipped and the jitter build run
    the same arithmetic in the same order. Both builds run in child processes (one library per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jit = os.path.join(root, '2g-gcn_amd', 'lib2ggcn_hip_jitter.so')
    if not os.path.exists(jit):   # the diagnostic build is not part of `make all`: build it here (hipcc is on the GPU box too)
        subprocess.run(['make', '-C', os.path.join(root, '2g-gcn_amd', 'csrc'), 'jitter', '-j4'], check=True, capture_output=True)
    ref = str(tmp_path / 'persist_ref.pt')
    tool = os.path.join(root, 'tools', 'persist_jitter_check.py')
    env = {k: v for k, v in os.environ.items() if k != 'TWOG_LIB_PATH'}
    r = subprocess.run([sys.executable, tool, 'write', ref, '1'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, tool, 'check', ref, '200'], env=dict(env, TWOG_LIB_PATH=jit), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert '0 tensors differ' in r.stdout, r.stdout[-500:]


def test_grouped_column_sums_equal_the_single_calls_bit_for_bit(K):
    """twog_colsum_n (the bias gradients of a backward stage in one pair of launches) against one twog_colsum call per
    problem: strided views, a row scale, ragged column counts (vector and scalar paths), accumulation into existing values,
    more than TWOG_COLSUM_MAX problems (two chunks)."""
    g = torch.Generator().manual_seed(5)
    base = torch.randn(20000, 700, generator=g).to(DEV)
    probs = []
    for i, (r0, r1, c0, c1, scaled, acc) in enumerate([(0, 20000, 0, 512, False, False), (5, 7001, 4, 260, True, True),
                                                       (0, 64, 0, 700, False, False), (100, 163, 1, 14, True, False),
                                                       (0, 15360, 64, 576, False, True)] * 4):
        x = base[r0:r1, c0:c1]
        rs = torch.randn(r1 - r0, generator=g).to(DEV) if scaled else None
        out0 = torch.randn(c1 - c0, generator=g).to(DEV)
        probs.append((x, rs, out0, acc))
    assert len(probs) > 16
    single = [K.colsum(x, rowscale=rs, out=o.clone(), accumulate=acc) for x, rs, o, acc in probs]
    many = [o.clone() for _, _, o, _ in probs]
    K.colsum_many([(x, rs, m, acc) for (x, rs, _, acc), m in zip(probs, many)])
    torch.cuda.synchronize()
    for a, b in zip(single, many):
        assert torch.equal(a, b)
    ref = (base[0:20000, 0:512].double()).sum(0)
    assert float((many[0].double() - ref).abs().max()) < 1e-3 * float(ref.abs().max())


# --------------------------------------------------------------------------------------------------------------- gates
@pytest.mark.parametrize('gs', [True, False])
def test_gates_filter_reorder_heads(K, gs):
    bs, T, E, h = 3, 6, 4, 32
    W = 5 * h
    x = rnd(bs * T * E, W, scale=0.5)
    cols = [0, h, 2 * h, 4 * h, 3 * h]
    w, b = rnd(1, 5 * h, seed=1, scale=0.2), rnd(1, seed=2)
    noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * (E + 2), bs, 2)) if gs else None

    def desc(dev):
        return dict(x=x.to(dev), seg_col=cols, hidden=h, w=w.to(dev), b=b.to(dev),
                    noise=noise.to(dev) if noise is not None else None, bs=bs, T=T, E=E, noise_entities=E + 2,
                    noise_offset=2, force_last=1, threshold=0.5)

    dc, dg = desc('cpu'), desc(DEV)
    hc, sc = F.gate_fwd(dc)
    hg, sg = K.gate_fwd(dg)
    close(sg, sc, rtol=1e-4, atol=1e-6, what='gate soft')
    safe = (sc - 0.5).abs() > 1e-4
    assert torch.equal(hg.cpu()[safe], hc[safe])
    assert torch.all(hg[:, T - 1] == 1)
    d_hard, d_soft, stm = rnd(bs, T, E, seed=3), rnd(bs, T, E, seed=4), (rnd(bs, T, E, seed=5) > 0).float()
    dg['p_save'], dg['soft'] = dc['p_save'].to(DEV), dc['soft'].to(DEV)
    lc = F.gate_bwd(dc, d_hard, d_soft, stm)
    lg = K.gate_bwd(dg, d_hard.to(DEV), d_soft.to(DEV), stm.to(DEV))
    close(lg, lc, rtol=1e-4, atol=1e-6, what='gate dlogit')
    lc2 = F.gate_bwd(dc, d_hard, None, None)
    lg2 = K.gate_bwd(dg, d_hard.to(DEV), None, None)
    close(lg2, lc2, rtol=1e-4, atol=1e-6, what='gate dlogit (no soft/mask)')
    # rank-1 update + weighted column sum on strided views
    dst = rnd(bs * T * E, W, seed=6)
    dst_g = dst.clone().to(DEV)
    F.rank1_update(dst[:, h:2 * h], lc, w.view(-1)[:h])
    K.rank1_update(dst_g[:, h:2 * h], lc.to(DEV), w.view(-1)[:h].contiguous().to(DEV))
    close(dst_g, dst, what='rank1')
    cs_c = F.colsum(x[:, h:3 * h], rowscale=lc)
    cs_g = K.colsum(x.to(DEV)[:, h:3 * h], rowscale=lc.to(DEV))
    close(cs_g, cs_c, rtol=1e-4, atol=1e-5, what='weighted colsum')
    big = rnd(20000, 48, seed=8)
    close(K.colsum(big.to(DEV)), F.colsum(big), rtol=1e-4, atol=1e-3, what='colsum tall')
    # filter
    soft = torch.rand(bs, T, E)
    for thr in (0.1, 0.5):
        fc = F.filter_fwd(soft, thr)
        fg = K.filter_fwd(soft.to(DEV), thr)
        assert torch.equal(fg[0].cpu(), fc[0]) and torch.equal(fg[1].cpu(), fc[1])
    # reorder
    hx = rnd(bs, T, E, 2 * h, seed=9)
    gate = (rnd(bs, T, E, seed=10) > 0.3).float()
    gate[0] = 0
    gate[1, :, 0] = 1
    assert torch.equal(K.reorder_fwd(hx.to(DEV), gate.to(DEV)).cpu(), F.reorder_fwd(hx, gate))
    close(K.reorder_bwd(hx.to(DEV), gate.to(DEV)), F.reorder_bwd(hx, gate), what='reorder bwd')
    # heads epilogue
    C = 13
    logits = rnd(bs * T * E, C, seed=11, scale=3)
    oc = F.logsoftmax_permute_fwd(logits, bs, T, E, C)
    og = K.logsoftmax_permute_fwd(logits.to(DEV), bs, T, E, C)
    close(og, oc, rtol=1e-5, atol=1e-5, what='logsoftmax permute')
    dout = rnd(bs, C, T, E, seed=12)
    close(K.logsoftmax_permute_bwd(oc.to(DEV), dout.to(DEV)), F.logsoftmax_permute_bwd(oc, dout), rtol=1e-4, atol=1e-5,
          what='logsoftmax bwd')
    # elementwise
    y, dy = rnd(500, 36, seed=13), rnd(500, 36, seed=14)
    close(K.relu_bwd(dy.to(DEV), y.to(DEV)), F.relu_bwd(dy, y), what='relu bwd')
    ybig, dybig = rnd(100, 80, seed=15), rnd(100, 80, seed=16)
    close(K.relu_bwd(dybig.to(DEV)[:, 3:40], ybig.to(DEV)[:, 3:40]), F.relu_bwd(dybig[:, 3:40], ybig[:, 3:40]),
          what='relu bwd unaligned view')
    a, bdst = rnd(64, 20, seed=17), rnd(64, 20, seed=18)
    bg = bdst.clone().to(DEV)
    F.add_rows(a, bdst)
    K.add_rows(a.to(DEV), bg)
    close(bg, bdst, what='add_rows')
    # adam
    p0, g0 = rnd(1000, seed=19), rnd(1000, seed=20)
    pc, mc, vc = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    pg, mg, vg = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in (1, 2, 3):
        F.adam_step(pc, g0, mc, vc, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
        K.adam_step(pg, g0.to(DEV), mg, vg, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
    close(pg, pc, rtol=1e-5, atol=1e-6, what='adam')
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    for _ in range(3):
        ref.grad = g0.clone()
        opt.step()
    close(pg, ref.data, rtol=1e-5, atol=1e-6, what='adam vs torch.optim.Adam')


def test_bn_fold_with_stats_reduce_hook(K):
    """sync-BN hook of distributed.DataParallel: reducing the (already complete) sums of one rank changes nothing, and
    doubling sums and frame count (two identical ranks) gives the same statistics."""
    N, bs, T, H = 26, 3, 5, 2
    xh = rnd(bs, T, H, 2048 + 4 * N).to(DEV)
    gamma, beta = (rnd(4 * N, seed=1).abs() + 0.5).to(DEV), rnd(4 * N, seed=2).to(DEV)

    def run(hook):
        rm, rv = torch.zeros(4 * N, device=DEV), torch.ones(4 * N, device=DEV)
        nbt = torch.tensor(0, dtype=torch.int64, device=DEV)
        ab, mi = K.bn_fold(xh, N, gamma, beta, rm, rv, nbt, True, stats_reduce=hook)
        return ab.cpu(), mi.cpu(), rm.cpu(), rv.cpu()

    base = run(None)
    same = run(lambda s, n: (s, n))
    two = run(lambda s, n: (s * 2, n * 2))
    for a, b in zip(base, same):
        close(b, a, rtol=1e-6, atol=1e-7, what='identity hook')
    for a, b in zip(base[:3], two[:3]):
        close(b, a, rtol=1e-6, atol=1e-7, what='two identical ranks')


def test_out_of_range_descriptors_are_rejected_before_any_launch(K):
    """Size limits of the C ABI (include/twog_gcn.h) return a negative code instead of launching; the device stays usable.
    Every call below fails its first range check, so the null data pointers are never dereferenced."""
    from twog_gcn_amd import _lib
    lib, st = K.lib, K._stream()
    max_nodes = lib.twog_gcn_max_nodes()
    assert max_nodes == 64
    assert lib.twog_bn_stats(None, 0, 8, max_nodes + 1, None, 4, st) < 0
    assert lib.twog_bn_stats(None, 0, 8, 0, None, 4, st) < 0
    assert lib.twog_gcn_attn2_fwd(None, None, 8, max_nodes + 1, None, None, st) < 0
    assert lib.twog_bigru_fwd(None, 5, 2, 3, 32, None, 0, st) < 0
    assert lib.twog_attn_fwd(None, 5, st) < 0                      # more descriptor groups than one launch carries
    g = _lib.Gate()
    g.bs, g.T, g.E, g.n_seg = 2, 3, 2, 9                             # more gate-input column segments than the struct holds
    assert lib.twog_gate_fwd(g, st) < 0
    # empty problems are no-ops, not errors
    assert lib.twog_gemm_f32(None, 0, 0, 0, None, 0, st) == 0
    assert lib.twog_gcn_attn2_fwd(None, None, 0, 34, None, None, st) == 0
    torch.cuda.synchronize()
    x = torch.ones(4, device=DEV)
    assert float((x + 1).sum()) == 8.0


# ------------------------------------------------------------------------------------------------ hipGraph loop cache
def _graph_collision_scenario():
    """Two BiGRU loops with different descriptors (buffers, sizes) alternate; with the hash cut to 0 bits they share one
    bucket, so only the descriptor-byte compare keeps them from replaying each other's captured pointers. Every buffer
    is allocated once and the C entry point is called directly, so each loop shows the SAME descriptor every time (the
    steady state of a training loop): first sighting = direct launches, second = capture, later ones = replays."""
    import ctypes
    from twog_gcn_amd import _lib as L
    K = twog_kernels.get_kernels()
    h, T = 32, 4
    cases = []
    for i, (bs, E) in enumerate(((3, 2), (5, 3))):
        d = dict(gi=rnd(bs, T, E, 6 * h, seed=i).to(DEV), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=0.2).to(DEV),
                 b_hh_f=rnd(3 * h, seed=20 + i).to(DEV), w_hh_r=rnd(3 * h, h, seed=30 + i, scale=0.2).to(DEV),
                 b_hh_r=rnd(3 * h, seed=40 + i).to(DEV))
        want = F.bigru_fwd([{k: v.cpu() for k, v in d.items()}], bs, T, h)[0][0]
        bufs = dict(out=torch.empty(bs, T, E, 2 * h, device=DEV), save=torch.empty(2, bs, T, E, 4 * h, device=DEV),
                    tmp=torch.empty(2, bs * E, 3 * h, device=DEV), zeros=torch.zeros(bs * E, h, device=DEV))
        arr = (L.BiGru * 1)()
        a = arr[0]
        a.gi, a.w_hh_f, a.b_hh_f, a.w_hh_r, a.b_hh_r = (d[k].data_ptr() for k in ('gi', 'w_hh_f', 'b_hh_f', 'w_hh_r', 'b_hh_r'))
        a.out, a.save, a.tmp_gh, a.zeros, a.E = (bufs['out'].data_ptr(), bufs['save'].data_ptr(), bufs['tmp'].data_ptr(),
                                                 bufs['zeros'].data_ptr(), E)
        cases.append((bs, arr, bufs, d, want))
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(5):            # sighting, capture, replays -- interleaved between the two loops
        for bs, arr, bufs, d, want in cases:
            bufs['out'].fill_(float('nan'))
            assert K.lib.twog_bigru_fwd(arr, 1, bs, T, h, *K.chain_workspace(DEV), st) == 0
            close(bufs['out'], want, rtol=1e-4, atol=1e-5, what=f'bigru rep {rep} bs {bs}')
    return K.graph_cache_stats()


def test_graph_cache_survives_hash_collisions():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import tests.test_kernels_gpu as t; n, c = t._graph_collision_scenario(); '
            'assert c > 0, (n, c); assert n >= 2, (n, c); print("collisions resolved:", n, c)')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, TWOG_GRAPH_HASH_BITS='0', TWOG_GRAPHS='1'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'collisions resolved' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------ position features / rare gate strategies
@pytest.mark.parametrize('periodic', [False, True])
def test_position_feature_kernels(K, periodic):
    bs, T, E, h, W = 3, 7, 4, 32, 96
    steps = torch.tensor([7.0, 5.0, 6.0])
    w, b = rnd(h, seed=1), rnd(h, seed=2)
    # time feature into a column block of wider rows
    rows_c, rows_g = torch.zeros(bs * T * E, W), torch.zeros(bs * T * E, W, device=DEV)
    s_c = F.pos_embed_fwd(rows_c[:, 32:64], bs, T, E, h, w=w, b=b, periodic=periodic, steps=steps, divide=not periodic)
    s_g = K.pos_embed_fwd(rows_g[:, 32:64], bs, T, E, h, w=w.to(DEV), b=b.to(DEV), periodic=periodic,
                          steps=steps.to(DEV), divide=not periodic)
    close(s_g, s_c, what='time scalars')
    close(rows_g, rows_c, rtol=2e-5, atol=2e-5, what='time embedding (other columns untouched)')
    # segment lengths from hard decisions, their embedding, and the backward pass through the scan
    u = (rnd(bs, T, E, seed=3) > 0).float()
    u[:, -1] = 1.0
    sl_c, sl_g = F.seglen_fwd(u, steps, not periodic), K.seglen_fwd(u.to(DEV), steps.to(DEV), not periodic)
    close(sl_g, sl_c, what='segment lengths')
    F.pos_embed_fwd(rows_c[:, 64:96], bs, T, E, h, w=w, b=b, periodic=periodic, s=sl_c.view(-1))
    K.pos_embed_fwd(rows_g[:, 64:96], bs, T, E, h, w=w.to(DEV), b=b.to(DEV), periodic=periodic, s=sl_g.view(-1))
    close(rows_g, rows_c, rtol=2e-5, atol=2e-5, what='segment-length embedding')
    ds = rnd(bs, T, E, seed=4)
    du_c, du_g = rnd(bs, T, E, seed=5), rnd(bs, T, E, seed=5).to(DEV)
    F.seglen_bwd(u, steps, not periodic, ds, du_c)
    K.seglen_bwd(u.to(DEV), steps.to(DEV), not periodic, ds.to(DEV), du_g)
    close(du_g, du_c, rtol=2e-5, atol=2e-5, what='d hard gates through the length scan')
    if periodic:
        dout = rnd(bs * T * E, W, seed=6)
        close(K.periodic_embed_bwd(dout.to(DEV)[:, 64:96], sl_g.view(-1)), F.periodic_embed_bwd(dout[:, 64:96], sl_c.view(-1)),
              rtol=2e-5, atol=2e-5, what='d scalar of the periodic embedding')
    # autograd check of the scan's backward rule on the CPU specification itself
    uu = u.clone().requires_grad_(True)
    acc, rels = torch.zeros(bs, E), []
    for t in range(T):
        xt = torch.full((bs, 1), float(t + 1)) / (steps.view(bs, 1) if not periodic else 1.0)
        rel = uu[:, t] * xt
        rel = torch.where(rel.bool(), rel - acc, rel)
        acc = acc + rel
        rels.append(rel)
    (torch.stack(rels, 1) * ds).sum().backward()
    close(du_c - rnd(bs, T, E, seed=5), uu.grad, rtol=1e-5, atol=1e-5, what='scan backward rule vs autograd')


def test_mul_and_scale_rows(K):
    a, b = rnd(5, 9, 3), rnd(5, 9, 3, seed=1)
    close(K.mul(a.to(DEV), b.to(DEV)), a * b, what='mul')
    o = rnd(5, 9, 3, seed=2)
    og = o.clone().to(DEV)
    K.mul(a.to(DEV), b.to(DEV), out=og, accumulate=True)
    close(og, o + a * b, what='mul accumulate')
    x = rnd(6, 40)
    s = rnd(6, seed=3)
    xg = x.clone().to(DEV)
    K.scale_rows(xg[:, 8:24], s.to(DEV))
    xc = x.clone()
    xc[:, 8:24] *= s.view(-1, 1)
    close(xg, xc, what='scale_rows on a column block')


@pytest.mark.parametrize('H,O,D,h,geo', [(2, 4, 64, 32, True),      # 6 entities: 15 pairs in 16 slots
                                         (1, 1, 32, 32, False),     # one pair
                                         (1, 5, 40, 16, True),      # fewer column pairs than threads
                                         (2, 7, 96, 48, True),      # 9 entities in the 10-entity variant
                                         (2, 8, 200, 64, False),    # 10 entities, a column count that is no power of two
                                         (2, 9, 128, 64, True)])    # 11 entities: the row-parallel (LDS) Gram
def test_entity_attention_streaming_regime(K, H, O, D, h, geo):
    """More than 1024 instances (the frame-level call of a real batch): the forward kernel whose Gram matrix is computed
    column-parallel from global memory (at most ten entities) and the backward kernel whose dL/dw is (at most 2 humans,
    8 objects), and their row-parallel fall-backs (11 entities / 9 objects)."""
    n_inst, ipc = 1100, 10
    dc = _attn_case('cpu', H, O, D, h, n_inst, ipc, geo, 1, seed=5)
    dg = _attn_case(DEV, H, O, D, h, n_inst, ipc, geo, 1, seed=5)
    if H == 1:
        for d in (dc, dg):
            d['msg_hh'] = None
            d.pop('out_hh')
    if O == 1:
        for d in (dc, dg):
            d['msg_oo'] = None
            d.pop('out_oo')
    F.attn_fwd([dc])
    K.attn_fwd([dg])
    close(dg['att'], dc['att'], rtol=1e-4, atol=1e-6, what='att weights')
    for k in dc:
        if k.startswith('out_'):
            close(dg[k], dc[k], rtol=1e-4, atol=1e-5, what=k)
    assert not torch.isnan(dg['att']).any()

    def bwd(dev, d, seed=200):
        t = lambda *s, sd=0: rnd(*s, seed=seed + sd).to(dev)
        natt = H * H + 2 * H * O + O * O
        b = dict(f=d, dfeat_accumulate=1, relu_mask_dmsg=1, dfeat_h=t(n_inst * H, D, sd=1), dfeat_o=t(n_inst * O, D, sd=2),
                 dw_extra=t(n_inst, natt, sd=3) if geo else None)
        for i, (rel, R) in enumerate((('hh', H), ('oh', H), ('ho', O), ('oo', O), ('so', O), ('sh', H))):
            if d.get('msg_' + rel) is None:
                continue
            b['dout_' + rel] = t(n_inst * R, h, sd=10 + i)
            S_ = {'hh': H, 'ho': H, 'oh': O, 'oo': O, 'so': 0, 'sh': 0}[rel]
            b['dmsg_' + rel] = torch.zeros(n_inst * S_ if S_ else n_inst, h, device=dev)
        return b

    dc['att'], dg['att'] = dc['att'].clone(), dc['att'].to(DEV)   # same saved weights on both sides
    bc, bg = bwd('cpu', dc), bwd(DEV, dg)
    F.attn_bwd([bc])
    K.attn_bwd([bg])
    for k in bc:
        if k.startswith('dmsg_') or k.startswith('dfeat_h') or k.startswith('dfeat_o'):
            close(bg[k], bc[k], rtol=2e-4, atol=2e-5, what=k)


@pytest.mark.parametrize('n_inst,ipc', [(240, 120), (1200, 120)])   # LDS-staged (latency) path and streaming path
def test_entity_attention_product_layout_full_size(K, n_inst, ipc):
    """The frame-level call exactly as ops.py lays it out at the BASELINE shape (H=2, O=8, h=512, T=120): features and
    received messages are column blocks of the entity rows, sender messages (and their gradients) column blocks of
    shared (rows, 2h) buffers, one geometry sender; a partly virtual clip and an all-virtual one."""
    H, O, h = 2, 8, 512
    D, Wh, Wo = 2 * h, 4 * h, 5 * h

    def build(dev):
        t = lambda *s, sd=0: rnd(*s, seed=40 + sd).to(dev)
        HUM, OBJ = t(n_inst * H, Wh, sd=1), t(n_inst * O, Wo, sd=2)
        MSGH, MSGO, MSGS = t(n_inst * H, 2 * h, sd=3), t(n_inst * O, 2 * h, sd=4), t(n_inst, h, sd=5)
        mask = torch.ones(n_inst // ipc, O)
        mask[0, O - 2:] = 0
        mask[-1] = 0
        f = dict(feat_h=HUM[:, :D], feat_o=OBJ[:, :D], msg_hh=MSGH[:, :h], msg_ho=MSGH[:, h:], msg_oh=MSGO[:, :h],
                 msg_oo=MSGO[:, h:], msg_so=MSGS, msg_sh=None, out_hh=HUM[:, 2 * h:3 * h], out_oh=HUM[:, 3 * h:],
                 out_ho=OBJ[:, 2 * h:3 * h], out_so=OBJ[:, 3 * h:4 * h], out_oo=OBJ[:, 4 * h:],
                 obj_mask=mask.to(dev), att=torch.zeros(n_inst, H * H + 2 * H * O + O * O, device=dev), n_inst=n_inst,
                 inst_per_clip=ipc, H=H, O=O, D=D, hidden=h, scale=1.0 / math.sqrt(D), recv_mask_ho=1)
        return f, HUM, OBJ

    fc, HUMc, OBJc = build('cpu')
    fg, HUMg, OBJg = build(DEV)
    F.attn_fwd([fc])
    K.attn_fwd([fg])
    close(fg['att'], fc['att'], rtol=1e-4, atol=1e-6, what='att')
    close(HUMg, HUMc, rtol=1e-4, atol=1e-5, what='human rows')
    close(OBJg, OBJc, rtol=1e-4, atol=1e-5, what='object rows')

    def bwd(dev, f):
        t = lambda *s, sd=0: rnd(*s, seed=70 + sd).to(dev)
        dHUM, dOBJ = t(n_inst * H, Wh, sd=1), t(n_inst * O, Wo, sd=2)
        dMSGH, dMSGO, dMSGS = (torch.zeros(n_inst * H, 2 * h, device=dev), torch.zeros(n_inst * O, 2 * h, device=dev),
                               torch.zeros(n_inst, h, device=dev))
        b = dict(f=f, dfeat_h=dHUM[:, :D], dfeat_o=dOBJ[:, :D], dfeat_accumulate=1, relu_mask_dmsg=1,
                 dmsg_hh=dMSGH[:, :h], dmsg_ho=dMSGH[:, h:], dmsg_oh=dMSGO[:, :h], dmsg_oo=dMSGO[:, h:], dmsg_so=dMSGS,
                 dout_hh=dHUM[:, 2 * h:3 * h], dout_oh=dHUM[:, 3 * h:], dout_ho=dOBJ[:, 2 * h:3 * h],
                 dout_so=dOBJ[:, 3 * h:4 * h], dout_oo=dOBJ[:, 4 * h:])
        return b, (dHUM, dOBJ, dMSGH, dMSGO, dMSGS)

    bc, outs_c = bwd('cpu', fc)
    bg, outs_g = bwd(DEV, fg)
    F.attn_bwd([bc])
    K.attn_bwd([bg])
    for name, g, c in zip(('dHUM', 'dOBJ', 'dMSGH', 'dMSGO', 'dMSGS'), outs_g, outs_c):
        close(g, c, rtol=2e-4, atol=2e-5, what=name)


# ------------------------------------------------------------------------------ general single-relation message passing
def _relation_case(dev, score, msg_mode, R, S, D, h, n_inst, ipc, exclude_self, send_mask, recv_mask, relu_scores, seed=0):
    t = lambda *s, sd=0, sc=1.0: rnd(*s, seed=seed + sd, scale=sc).to(dev)
    W = 2 * h + 8
    d = dict(score_mode=score, msg_mode=msg_mode, n_inst=n_inst, inst_per_clip=ipc, R=R, S=S, D=D, hidden=h,
             exclude_self=exclude_self, relu_scores=relu_scores, scale=1.0 / math.sqrt(D),
             out=torch.zeros(n_inst * R, W, device=dev)[:, 4:4 + h])
    n_clip = n_inst // ipc
    if send_mask:
        m = (rnd(n_clip, S, seed=seed + 1) > -0.4).float()
        m[0] = 0.0            # a clip whose senders are all virtual: weights must be 0, not NaN
        d['send_mask'] = m.to(dev)
    if recv_mask:
        d['recv_mask'] = (rnd(n_clip, R, seed=seed + 2) > -0.4).float().to(dev)
    if score == F.REL_DOT:
        d.update(q=t(n_inst * R, D + 4, sd=3)[:, :D], k=t(n_inst, S, D, sd=4))     # 2-D strided and 3-D row sets
        if relu_scores:
            d['score_bias'] = t(1, sd=5)
    elif score == F.REL_ADDITIVE:
        d.update(a_r=t(n_inst * R, sd=6), c_s=t(n_inst * S, sd=7))
    elif score == F.REL_DISTANCE:
        dist = rnd(n_inst, S, R, seed=seed + 8).abs() + 0.05
        dist[rnd(n_inst, S, R, seed=seed + 9) > 1.0] = 0.0                         # distance 0 = "no such sender"
        d['dist'] = dist.to(dev).transpose(1, 2)                                    # strided (n_inst, R, S) view
    if msg_mode == F.REL_MSG_SENDER:
        d['msg'] = t(n_inst * S, 2 * h, sd=10)[:, h:]
    else:
        d.update(p_r=t(n_inst * R, h, sd=11), p_s=t(n_inst * S, h, sd=12))
    return d


@pytest.mark.parametrize('score,msg_mode,R,S,excl,smask,rmask,relu', [
    (F.REL_SUM, F.REL_MSG_PAIR, 2, 4, 0, 1, 0, 0),        # relational objects -> human
    (F.REL_SUM, F.REL_MSG_PAIR, 5, 5, 1, 1, 0, 0),        # relational objects -> object (self excluded)
    (F.REL_SUM, F.REL_MSG_SENDER, 4, 1, 0, 0, 1, 0),      # geometry -> objects, receiver mask
    (F.REL_DOT, F.REL_MSG_PAIR, 2, 2, 1, 0, 0, 0),        # specific + dot, humans -> human
    (F.REL_DOT, F.REL_MSG_SENDER, 4, 2, 0, 0, 1, 1),      # general (bilinear: relu + bias), humans -> objects
    (F.REL_ADDITIVE, F.REL_MSG_SENDER, 2, 9, 0, 1, 0, 0),  # concat, objects -> human
    (F.REL_ADDITIVE, F.REL_MSG_PAIR, 9, 9, 1, 1, 0, 0),    # concat + specific, objects -> object
    (F.REL_DISTANCE, F.REL_MSG_SENDER, 3, 6, 0, 1, 0, 0),
    (F.REL_MEAN, F.REL_MSG_PAIR, 6, 6, 1, 1, 0, 0)])
def test_relation_kernels(K, score, msg_mode, R, S, excl, smask, rmask, relu):
    D, h, n_inst, ipc = 24, 40, 6, 3
    dc = _relation_case('cpu', score, msg_mode, R, S, D, h, n_inst, ipc, excl, smask, rmask, relu)
    dg = _relation_case(DEV, score, msg_mode, R, S, D, h, n_inst, ipc, excl, smask, rmask, relu)
    dc['att'], dg['att'] = torch.zeros(n_inst, R, S), torch.zeros(n_inst, R, S, device=DEV)
    F.relation_fwd(dc)
    K.relation_fwd(dg)
    close(dg['att'], dc['att'], rtol=1e-4, atol=1e-6, what='weights')
    assert not torch.isnan(dg['att']).any()
    close(dg['out'], dc['out'], rtol=1e-4, atol=1e-5, what='out')

    def bwd(dev, d):
        t = lambda *s, sd=0: rnd(*s, seed=200 + sd).to(dev)
        b = dict(f=d, dout=t(n_inst * R, h, sd=1), relu_mask_dmsg=1)
        if msg_mode == F.REL_MSG_SENDER:
            b['dmsg'] = torch.zeros(n_inst * S, h, device=dev)
        else:
            b.update(dp_r=torch.zeros(n_inst * R, h, device=dev), dp_s=torch.zeros(n_inst * S, h, device=dev))
        if score == F.REL_DOT:
            b.update(dq=t(n_inst * R, D, sd=2), dq_accumulate=1, dk=torch.zeros(n_inst * S, D, device=dev),
                     dscore_sum=torch.zeros(n_inst, device=dev) if relu else None)
        elif score == F.REL_ADDITIVE:
            b.update(da_r=torch.zeros(n_inst * R, device=dev), dc_s=torch.zeros(n_inst * S, device=dev))
        return b

    bc, bg = bwd('cpu', dc), bwd(DEV, dg)
    F.relation_bwd(bc)
    K.relation_bwd(bg)
    for k in ('dmsg', 'dp_r', 'dp_s', 'dq', 'dk', 'da_r', 'dc_s'):
        if bc.get(k) is not None:
            close(bg[k], bc[k], rtol=2e-4, atol=2e-5, what=k)
    if bc.get('dscore_sum') is not None:   # per-instance partials on the device, their total in the specification
        close(bg['dscore_sum'].sum(), bc['dscore_sum'].sum(), rtol=2e-4, atol=2e-5, what='d score bias')


def test_relation_self_relation_accumulates_both_feature_gradients(K):
    """humans -> human with dot scores: queries and keys are the SAME rows and both gradients are added into one
    buffer (ops.py passes dq = dk = the entity-row gradient columns)."""
    R = S = 3
    D, h, n_inst = 16, 8, 4
    feat = rnd(n_inst * R, D, seed=1)
    msg = rnd(n_inst * S, h, seed=2)
    dout = rnd(n_inst * R, h, seed=3)

    def run(Kx, dev):
        f_ = feat.to(dev)
        d = dict(score_mode=F.REL_DOT, msg_mode=F.REL_MSG_SENDER, n_inst=n_inst, inst_per_clip=1, R=R, S=S, D=D, hidden=h,
                 exclude_self=1, scale=0.25, q=f_, k=f_, msg=msg.to(dev), out=torch.zeros(n_inst * R, h, device=dev))
        Kx.relation_fwd(d)
        dfeat = torch.ones(n_inst * R, D, device=dev)
        Kx.relation_bwd(dict(f=d, dout=dout.to(dev), dmsg=torch.zeros(n_inst * S, h, device=dev), dq=dfeat, dk=dfeat,
                             dq_accumulate=1, dk_accumulate=1))
        return d['out'], dfeat

    (oc, gc), (og, gg) = run(F, 'cpu'), run(K, DEV)
    close(og, oc, what='out')
    close(gg, gc, rtol=2e-4, atol=2e-5, what='dq + dk into one buffer')


# ------------------------------------------------------------------------------------ sender-side projection glue
@pytest.mark.parametrize('H,O,ph_on,ps_on', [(2, 8, True, True), (1, 5, True, False), (2, 4, False, True), (3, 9, True, True)])
def test_sender_side_projection_kernels(K, H, O, ph_on, ps_on):
    n_inst, ipc, cols = 12, 4, 96
    natt = H * H + 2 * H * O + O * O
    off = H * H + H * O
    att = torch.softmax(rnd(n_inst, natt, seed=1), -1)
    mask = (rnd(n_inst // ipc, O, seed=2) > -0.5).float()
    mask[0] = 0
    gi = rnd(n_inst * O, cols, seed=3)
    ph = rnd(n_inst * H, cols, seed=4) if ph_on else None
    ps = rnd(n_inst, cols, seed=5) if ps_on else None
    dv = lambda t: None if t is None else t.to(DEV)
    gc, gg = gi.clone(), gi.clone().to(DEV)
    F.ssp_fwd(gc, ph, ps, att, mask, n_inst, ipc, H, O, off)
    K.ssp_fwd(gg, dv(ph), dv(ps), att.to(DEV), mask.to(DEV), n_inst, ipc, H, O, off)
    close(gg, gc, what='ssp fwd')
    # the scatter equals projecting the aggregated messages: gi += mask * (att @ ph + ps)
    dgi = rnd(n_inst * O, cols, seed=6)
    dwc, dwg = torch.zeros(n_inst, natt), torch.zeros(n_inst, natt, device=DEV)
    qhc, qsc = F.ssp_bwd(dgi, ph, att, mask, n_inst, ipc, H, O, off, ps_on, dw=dwc)
    qhg, qsg = K.ssp_bwd(dgi.to(DEV), dv(ph), att.to(DEV), mask.to(DEV), n_inst, ipc, H, O, off, ps_on, dw=dwg)
    if ph_on:
        close(qhg, qhc, what='qh')
        close(dwg, dwc, rtol=1e-4, atol=1e-5, what='dw')
    if ps_on:
        close(qsg, qsc, what='qs')
    # adjointness: <ssp_fwd(0; ph, ps), dgi> == <ph, qh> + <ps, qs>
    z = torch.zeros(n_inst * O, cols)
    F.ssp_fwd(z, ph, ps, att, mask, n_inst, ipc, H, O, off)
    lhs = (z * dgi).sum()
    rhs = ((ph * qhc).sum() if ph_on else 0.0) + ((ps * qsc).sum() if ps_on else 0.0)
    assert abs(float(lhs - rhs)) < 1e-3 * max(1.0, abs(float(lhs)))


def test_attention_backward_takes_extra_weight_gradient(K):
    H, O, D, h, n_inst, ipc = 2, 4, 64, 32, 6, 3
    dc = _attn_case('cpu', H, O, D, h, n_inst, ipc, True, 1)
    dg = _attn_case(DEV, H, O, D, h, n_inst, ipc, True, 1)
    F.attn_fwd([dc])
    K.attn_fwd([dg])
    natt = H * H + 2 * H * O + O * O
    extra = rnd(n_inst, natt, seed=77)

    def bwd(dev, d):
        t = lambda *s, sd=0: rnd(*s, seed=100 + sd).to(dev)
        b = dict(f=d, dfeat_accumulate=0, relu_mask_dmsg=1, dfeat_h=torch.zeros(n_inst * H, D, device=dev),
                 dfeat_o=torch.zeros(n_inst * O, D, device=dev), dw_extra=extra.to(dev))
        for i, (rel, R) in enumerate((('hh', H), ('oh', H), ('ho', O), ('oo', O), ('so', O), ('sh', H))):
            b['dout_' + rel] = t(n_inst * R, h, sd=10 + i)
            S_ = {'hh': H, 'ho': H, 'oh': O, 'oo': O, 'so': 0, 'sh': 0}[rel]
            b['dmsg_' + rel] = torch.zeros(n_inst * S_ if S_ else n_inst, h, device=dev)
        return b

    bc, bg = bwd('cpu', dc), bwd(DEV, dg)
    F.attn_bwd([bc])
    K.attn_bwd([bg])
    for k in ('dfeat_h', 'dfeat_o', 'dmsg_ho', 'dmsg_oo'):
        close(bg[k], bc[k], rtol=2e-4, atol=2e-5, what=k)
    b0 = bwd('cpu', dc)
    b0['dw_extra'] = None
    F.attn_bwd([b0])
    assert float((b0['dfeat_h'] - bc['dfeat_h']).abs().max()) > 1e-4   # the extra term does reach the features


def test_bigru_forward_graph_capture_and_replay_at_baseline_width(K, monkeypatch):
    """BASELINE-size BiGRU step (three entity types, 528 tiles): first sighting (direct launches), capture into a
    hipGraph and two replays must all give the specification's result (the opt-in graph path: TWOG_GRAPHS=1)."""
    from twog_gcn_amd import _lib as L
    monkeypatch.setenv('TWOG_GRAPHS', '1')
    bs, T, h = 64, 3, 512
    ws = 0.2 * math.sqrt(64.0 / h)
    arr = (L.BiGru * 3)()
    keep, wants = [], []
    for i, E in enumerate((2, 8, 1)):
        d = dict(gi=rnd(bs, T, E, 6 * h, seed=i), w_hh_f=rnd(3 * h, h, seed=10 + i, scale=ws), b_hh_f=rnd(3 * h, seed=20 + i),
                 w_hh_r=rnd(3 * h, h, seed=30 + i, scale=ws), b_hh_r=rnd(3 * h, seed=40 + i))
        wants.append(F.bigru_fwd([d], bs, T, h)[0][0])
        g = {k: v.to(DEV) for k, v in d.items()}
        bufs = dict(out=torch.empty(bs, T, E, 2 * h, device=DEV), save=torch.empty(2, bs, T, E, 4 * h, device=DEV),
                    tmp=torch.empty(2, bs * E, 3 * h, device=DEV), zeros=torch.zeros(bs * E, h, device=DEV))
        a = arr[i]
        a.gi, a.w_hh_f, a.b_hh_f, a.w_hh_r, a.b_hh_r = (g[k].data_ptr() for k in ('gi', 'w_hh_f', 'b_hh_f', 'w_hh_r', 'b_hh_r'))
        a.out, a.save, a.tmp_gh, a.zeros, a.E = (bufs['out'].data_ptr(), bufs['save'].data_ptr(), bufs['tmp'].data_ptr(),
                                                 bufs['zeros'].data_ptr(), E)
        keep.append((g, bufs))
    st = torch.cuda.current_stream().cuda_stream
    n0 = K.graph_cache_stats()[0]
    for rep in range(4):
        for _, bufs in keep:
            bufs['out'].fill_(float('nan'))
        assert K.lib.twog_bigru_fwd(arr, 3, bs, T, h, *K.chain_workspace(DEV), st) == 0
        for (_, bufs), want in zip(keep, wants):
            close(bufs['out'], want, rtol=1e-4, atol=1e-5, what=f'bigru rep {rep}')
    assert K.graph_cache_stats()[0] == n0 + 1   # the loop was captured (and replayed twice)


@pytest.mark.parametrize('bkm', [False, True])
def test_gemm_ksplit_class(K, bkm):
    """Few 64x64 tiles with a reduction worth splitting (the recurrent chains' launches): 8-wave workgroups whose two
    wave groups take alternate k-chunks and add their partial tiles through LDS. Checked against fp64; the class bit is
    asserted, as for the 128x128 class."""
    for (M, N, K_, bias, act, acc) in ((640, 1024, 512, True, 1, False), (1280, 512, 1536, False, 0, True),
                                      (130, 200, 288, True, 0, True),
                                      # at most 256 tiles of 32 x 64: the 32-row variant of small batches (8 clips)
                                      (176, 512, 1536, False, 0, True), (48, 1536, 512, True, 1, False), (33, 70, 256, True, 0, False)):
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(M, K_, generator=g).to(DEV)
        B = ((torch.randn(K_, N, generator=g) if bkm else torch.randn(N, K_, generator=g)) * 0.1).to(DEV)
        b = torch.randn(N, generator=g).to(DEV) if bias else None
        C0 = torch.randn(M, N, generator=g).to(DEV)
        Cg = C0.clone()
        # (no split-K workspace: the form the time loops use)
        K.gemm([dict(A=A, B=B, C=Cg, bias=b, act=act, accumulate=acc)], b_kmajor=bkm, split_k_workspace=False)
        cls = K.gemm_last_class()
        assert cls & K.GEMM_KSPLIT and not cls & K.GEMM_TILE128, hex(cls)
        assert bool(cls & K.GEMM_ROWS32) == (-(-M // 32) * -(-N // 64) <= 256), (hex(cls), M, N)
        ref = A.double() @ (B.double() if bkm else B.double().t())
        if bias:
            ref = ref + b.double()
        if acc:
            ref = ref + C0.double()
        if act:
            ref = torch.relu(ref)
        assert (Cg.double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize('bkm', [False, True])
@pytest.mark.parametrize('force', ['0', '2', '3', '8'])
def test_gemm_chain_split_over_workgroups(K, bkm, force):
    """Chain launches with the reduction split over WORKGROUPS and combined inside the launch by the last arriver
    (TWOG_GEMM_CLASS_XSPLIT): 64x64 and 32x64 tiles, bias / ReLU / accumulate epilogues, ragged edges, a grouped launch
    that mixes reduction lengths, the library's own choice of the slice count (force 0) and forced ones -- in a child
    process per setting (the variable is read once). Against fp64; bit-identical from launch to launch (the partials are
    added in slice order whoever arrives last); the tickets return to zero."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels(); DEV = 'cuda:0'; bkm = %r; force = %r
def run(probs, expect_xs):
    outs = []
    for rep in range(3):
        ps = [dict(p, C=p['C0'].clone()) for p in probs]
        K.gemm([{k: v for k, v in p.items() if k != 'C0'} for p in ps], b_kmajor=bkm, chain=True)
        cls = K.gemm_last_class()
        outs.append([p['C'] for p in ps])
    assert bool(cls & K.GEMM_XSPLIT) == expect_xs, (hex(cls), expect_xs, [tuple(p['C'].shape) for p in ps])
    for p, Cg in zip(probs, outs[0]):
        A, B = p['A'], p['B']
        ref = A.double() @ (B.double() if bkm else B.double().t())
        if p.get('bias') is not None: ref = ref + p['bias'].double()
        if p.get('accumulate'): ref = ref + p['C0'].double()
        if p.get('act'): ref = torch.relu(ref)
        err = (Cg.double() - ref).abs().max().item()
        assert err <= 3e-5 * ref.abs().max().item(), (err, tuple(Cg.shape))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b), 'launch-to-launch difference'
    ptr, nbytes = K.chain_workspace(DEV)
    tickets = K._ws[(str(DEV), 'chain', int(K._stream() or 0))][:4096].view(torch.int32)
    assert int(tickets.abs().max()) == 0, 'tickets not returned to zero'
def prob(M, N, K_, bias, act, acc, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K_, generator=g).to(DEV)
    B = ((torch.randn(K_, N, generator=g) if bkm else torch.randn(N, K_, generator=g)) * 0.1).to(DEV)
    return dict(A=A, B=B, C0=torch.randn(M, N, generator=g).to(DEV), bias=torch.randn(N, generator=g).to(DEV) if bias else None,
                act=act, accumulate=acc)
forced = force != '0'
def auto(probs):   # the library's own decision for this launch
    K.gemm([dict({k: v for k, v in p.items() if k != 'C0'}, C=p['C0'].clone()) for p in probs], b_kmajor=bkm, chain=True)
    return bool(K.gemm_last_class() & K.GEMM_XSPLIT)
# (shape, must the library's own model split it? None = either way)
for (M, N, K_, bias, act, acc), model_xs in (((1408, 512, 1536, False, 0, True), None),     # BiGRU backward carry, bs64: 176 tiles
                                             ((1280, 1024, 1536, False, 0, False), True),   # segment d_mg, bs64: 320 tiles
                                             ((176, 512, 1536, False, 0, True), True),      # 8 clips: 32-row tiles
                                             ((176, 1536, 512, True, 1, False), True),      # 8 clips: W_hh projection
                                             ((130, 200, 288, True, 0, True), None),        # ragged edges, 9 k-tiles
                                             ((33, 70, 256, True, 1, False), None)):
    if force == '8' and K_ < 512:
        continue
    p = [prob(M, N, K_, bias, act, acc, M + N)]
    expect = True if forced else (model_xs if model_xs is not None else auto(p))
    run(p, expect)
# grouped launch mixing reduction lengths (the segment level's projection step: K = h and K = 2h)
probs = [prob(128, 1536, 512, True, 0, False, 1), prob(128, 1536, 1024, False, 0, False, 2),
         prob(64, 1536, 512, True, 0, False, 3), prob(64, 1536, 1024, False, 0, True, 4)]
run(probs, True if forced else auto(probs))
print('OK')
""" % (ROOT, bkm, force)
    env = dict(os.environ, TWOG_GEMM_XSPLIT=force, TWOG_X3_ROWS128='0')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_gemm_x3_error_equals_the_native_fp32_mfma_kernels():
    """The 128x128 class -- and the 64x64 class of the recurrent-chain launches with K >= 256 -- multiply on the bf16 matrix
    cores after an EXACT three-way split of every fp32 operand element (csrc/gemm_f32.hip, X3; default). In a child process per mode (the switch is read once): the four operand layouts,
    split-K, bias / ReLU / accumulate, grouped k-major rows, values over 60 orders of magnitude, exact zeros and a
    denormal -- the class bit is asserted, the error against an fp64 product must stay within 1.25 x the native fp32-MFMA
    kernels' error on the same inputs (+ 2e-7 of the largest output), and the result is bit-identical from launch to
    launch."""
    code = r"""
import sys, json, torch
sys.path.insert(0, %r)
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels(); DEV = 'cuda:0'
res = {}
g = torch.Generator().manual_seed(7)
def case(name, M, N, Kk, akm, bkm, bias=False, act=0, acc=False, scale_a=None, grouped=False, chain=False):
    A = torch.randn((Kk, M) if akm else (M, Kk), generator=g)
    if scale_a is not None:   # per-element magnitudes over many decades (the split must be exact at every exponent)
        A = A * 10.0 ** torch.randint(-scale_a, scale_a, A.shape, generator=g).float()
        A.view(-1)[::97] = 0.0
        A.view(-1)[5] = 1e-39
    B = torch.randn((Kk, N) if bkm else (N, Kk), generator=g) * 0.1
    A, B = A.to(DEV), B.to(DEV)
    if grouped:   # k-major operand whose rows are (outer, inner) grouped: all but the first of every 5 rows
        full = torch.randn(Kk // 4 * 5, M, generator=g).to(DEV)
        A = full.view(Kk // 4, 5, M)[:, 1:, :]
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    C0 = torch.randn(M, N, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        C = C0.clone()
        K.gemm([dict(A=A, B=B, C=C, bias=b, act=act, accumulate=acc)], a_kmajor=akm, b_kmajor=bkm, chain=chain,
               split_k_workspace=not chain)
        outs.append(C)
    cls = K.gemm_last_class()
    assert bool(cls & K.GEMM_TILE128) != chain, (name, hex(cls))
    assert torch.equal(outs[0], outs[1]), name
    Ad = (A.reshape(-1, M).t() if akm else A).double()
    ref = Ad @ (B.double() if bkm else B.double().t())
    if bias: ref = ref + b.double()
    if acc: ref = ref + C0.double()
    if act: ref = torch.relu(ref)
    res[name] = dict(err=float((outs[0].double() - ref).abs().max() / ref.abs().max()), cls=cls)
case('NN', 4096, 1536, 512, False, False, bias=True, act=1)
case('NT', 4096, 512, 1536, False, True, acc=True)
case('TT splitk', 1536, 512, 16384, True, True)
case('TN', 1024, 2048, 1024, True, False, bias=True)
case('NN wide magnitudes', 8192, 1024, 1024, False, False, scale_a=30)
case('TT grouped rows', 1536, 512, 8192, True, True, grouped=True)
# the 64x64 class of the recurrent chains (row-major A; 8 waves with the k-split inside the workgroup, and 4 waves)
case('chain NT carry', 1408, 512, 1536, False, True, acc=True, chain=True)
case('chain NN sender MLPs', 1280, 1024, 512, False, False, bias=True, act=1, chain=True)
case('chain NT wide magnitudes', 1280, 1024, 1536, False, True, scale_a=30, chain=True)
print('RES ' + json.dumps(res))
""" % (ROOT,)
    out = {}
    for mode in ('1', '0'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TWOG_GEMM_X3=mode, TWOG_GEMM_XSPLIT='1', TWOG_X3_ROWS128='0'),
                           capture_output=True, text=True, timeout=900)   # XSPLIT=1: no split over workgroups (fp32 kernels);
                                                                          # ROWS128=0: like with like (the 128-row chain tiles sum k in ONE wave group where the native kernels use two: their own test)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        out[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('RES ')][0][4:])
    X3 = twog_kernels.get_kernels().GEMM_X3
    for name, x in out['1'].items():
        n = out['0'][name]
        assert x['cls'] & X3 and not n['cls'] & X3, (name, hex(x['cls']), hex(n['cls']))
        assert x['err'] <= 1.25 * n['err'] + 2e-7, (name, x['err'], n['err'])
        assert x['err'] < 3e-5, (name, x['err'])
    print({k: (f"{v['err']:.2e}", f"{out['0'][k]['err']:.2e}") for k, v in out['1'].items()})


# |mean signed relative error| allowed on SAME-SIGN operands (see the test below): the 64x64 chain class keeps the five small
# chunk products in a second accumulator; the 128x128 class adds all six into the running sum (measured -4.3e-7 / -6.5e-7 /
# -2.07e-6 at K = 1 536 / 2 048 / 61 440 in eight slabs: 1.7e-9 per MFMA accumulation, relative to the accumulator)
X3_SAME_SIGN_BIAS_CAP = {'64': 1e-7, '128': 1.0e-6, '128 K=61440': 3.0e-6}


def test_gemm_x3_same_sign_and_wide_exponent_operands():
    """Operands chosen AGAINST the X3 scheme (tools/x3_bias_probe.py): post-ReLU activations x non-negative values with K
    up to 61 440 (the dW reductions), post-ReLU x signed weights, a 2^40 exponent spread inside every dot product -- in
    both tile classes. On same-sign operands every rounding that is not round-to-nearest shows as a BIAS. What round 4
    measured with this probe (profiles/r04_x3_products_6_vs_8_before_fix.txt): the three dropped chunk products are NOT the
    issue (a build with 8 products has the same numbers to three digits); the bf16 MFMA's accumulate is: every
    v_mfma_f32_32x32x16_bf16 that adds into a LARGE accumulator loses ~1.7e-9 of it (towards zero), where the fp32 MFMA rounds
    to nearest (bias 4e-10 on the same data). The 64x64 class keeps the five small products in a second accumulator (one
    accumulation into the main one per 16 k: -4e-8 at K = 1 536); the 128x128 class has no registers for that at two
    workgroups per CU and adds all six into the running sum (-4.3e-7 at K = 1 536, -2.1e-6 at K = 61 440); the fix that
    chains each k-step through a fresh accumulator (TWOG_X3_TMPACC=1) exists at build time and spills (profiles/HISTORY.md section 8).
    On SIGNED operands -- every GEMM of this model multiplies by signed weights or signed gradients -- the bias is 3e-9.
    Required: the class bit (X3 ran); the same-sign bias within the caps above (the documented state, so that a change for
    the worse fails); signed cases: |mean| <= 1e-7 of sum |a b|; max error <= 3e-6 of sum |a b| everywhere."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'x3_bias_probe.py'), '--json'],
                       env=dict(os.environ, TWOG_GEMM_XSPLIT='1'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rows = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('ROWS ')][0][5:])
    assert len(rows) >= 9
    print({r_['case']: (f"{r_['mean_err_over_sum_abs']:+.1e}", f"{r_['max']:.1e}") for r_ in rows})
    for row in rows:
        assert row['x3'], row
        assert row['max'] <= 3e-6, row
        if 'mean_rel_err_same_sign' in row:
            key = '64' if not row['tile128'] else ('128 K=61440' if '61440' in row['case'] else '128')
            assert abs(row['mean_rel_err_same_sign']) <= X3_SAME_SIGN_BIAS_CAP[key], row
            assert row['mean_rel_err_same_sign'] <= 0.0, ('the bias is towards zero', row)
        else:
            assert abs(row['mean_err_over_sum_abs']) <= 1e-7, row


def test_gemm_x3_dw_split_accumulator_option_removes_the_same_sign_bias(K):
    """TWOG_X3_DW_SPLIT_ACC=1 (off by default: the dW launches cost +18 ... 25 %, profiles/
    r05_dw_split_accumulator.txt): the TT launches of the 128x128 class chain every k-step's six products through a fresh
    accumulator and add it to the running sum with fp32 VALU adds, one workgroup per CU. On the same-sign K = 61 440 case the
    bias falls from -2.1e-6 to the native fp32-MFMA kernel's level: required |bias| <= 1e-7 and rms <= 2.2e-7 (2 x the fp32-MFMA
    kernel's 1.07e-7); the signed cases keep their bounds; a launch with a column-sum request agrees with fp64."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'x3_bias_probe.py'), '--json'],
                       env=dict(os.environ, TWOG_GEMM_XSPLIT='1', TWOG_X3_DW_SPLIT_ACC='1'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rows = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('ROWS ')][0][5:])
    tt = [row for row in rows if ' TT ' in row['case']]
    assert len(tt) >= 3
    for row in tt:
        assert row['x3'] and row['tile128'], row
        assert row['max'] <= 1e-6, row
        if 'mean_rel_err_same_sign' in row:
            assert abs(row['mean_rel_err_same_sign']) <= 1e-7 and row['rms'] <= 2.2e-7, row
        else:
            assert abs(row['mean_err_over_sum_abs']) <= 1e-7, row
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels(); g = torch.Generator().manual_seed(5)
A = (torch.randn(30720, 640, generator=g) + 0.25).cuda(); B = (torch.randn(30720, 512, generator=g) * 0.1).cuda()
C, cs = torch.empty(640, 512, device='cuda'), torch.empty(640, device='cuda')
K.gemm([dict(A=A, B=B, C=C, colsum=cs)], a_kmajor=True, b_kmajor=True)
ref, rcs = A.double().t() @ B.double(), A.double().sum(0)
assert float((C.double() - ref).abs().max() / ref.abs().max()) <= 1e-6
assert float((cs.double() - rcs).abs().max() / rcs.abs().max()) <= 3e-6
print('OK')
""" % (ROOT,)
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TWOG_X3_DW_SPLIT_ACC='1'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_gemm_x3_nonfinite_operands_stay_nonfinite(K):
    """Documented difference of the X3 kernels (csrc/gemm_f32.hip, split3): an operand of +-Inf gives NaN where an fp32
    multiply gives +-Inf (the split subtracts Inf - Inf); a NaN stays a NaN. Either way the affected outputs are NOT finite
    -- a broken run cannot look healthy -- and every other output is untouched."""
    g = torch.Generator().manual_seed(3)
    M, N, Kk = 2048, 2048, 512   # 256 tiles of 128 x 128: the X3 class
    A = torch.randn(M, Kk, generator=g)
    B = torch.randn(N, Kk, generator=g) * 0.1
    A[5, 17] = float('inf')
    A[9, 400] = float('nan')
    B[33, 3] = float('-inf')
    C = torch.zeros(M, N, device=DEV)
    K.gemm([dict(A=A.to(DEV), B=B.to(DEV), C=C, bias=None, act=0, accumulate=False)])
    assert K.gemm_last_class() & K.GEMM_X3
    bad = torch.zeros(M, N, dtype=torch.bool)
    bad[5, :] = True; bad[9, :] = True; bad[:, 33] = True
    Cc = C.cpu()
    assert not torch.isfinite(Cc[bad]).any(), 'an output fed by Inf / NaN came out finite'
    assert torch.isfinite(Cc[~bad]).all()
    ref = (torch.nan_to_num(A, nan=0.0, posinf=0.0, neginf=0.0).double() @ torch.nan_to_num(B, nan=0.0, posinf=0.0, neginf=0.0).double().t())
    assert (Cc[~bad].double() - ref[~bad]).abs().max() <= 3e-5 * ref[~bad].abs().max()


def test_gemm_x3_chain_k_tiles_per_barrier_is_bit_identical():
    """The X3 chain kernels take two k-tiles per barrier interval where every reduction allows it (TWOG_X3S_KU, default 2;
    csrc/gemm_f32.hip::gemm_mainloop_x3s): the same MFMA sequence into the same accumulators, so the results are the
    one-k-tile-per-barrier kernels' bit for bit -- plain and gate-fused launches, 8-wave and 4-wave tiles."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels(); DEV = 'cuda:0'
g = torch.Generator().manual_seed(11)
outs = []
for (M, N, Kk, bkm, acc) in ((1408, 512, 1536, True, True), (1280, 512, 1024, True, True), (1280, 1024, 512, False, False),
                             (1920, 1024, 1536, True, False), (1408, 1536, 512, False, False)):
    A = torch.randn(M, Kk, generator=g).to(DEV)
    B = (torch.randn((Kk, N) if bkm else (N, Kk), generator=g) * 0.1).to(DEV)
    C = torch.randn(M, N, generator=g).to(DEV)
    K.gemm([dict(A=A, B=B, C=C, bias=None, act=0, accumulate=acc)], b_kmajor=bkm, chain=True)
    assert K.gemm_last_class() & K.GEMM_X3
    outs.append(C.cpu())
# the BiGRU chains (fused forward step, gate-fused backward carry)
h, bs, T = 512, 64, 3
types = []
for i, E in enumerate((2, 8, 1)):
    types.append(dict(gi=torch.randn(bs, T, E, 6 * h, generator=g).to(DEV), w_hh_f=(torch.randn(3 * h, h, generator=g) * 0.07).to(DEV),
                      b_hh_f=torch.randn(3 * h, generator=g).to(DEV), w_hh_r=(torch.randn(3 * h, h, generator=g) * 0.07).to(DEV),
                      b_hh_r=torch.randn(3 * h, generator=g).to(DEV)))
res = K.bigru_fwd(types, bs, T, h)
bt = [dict(d_out=torch.randn(o.shape, generator=g).to(DEV), save=sv, out=o, w_hh_f=d['w_hh_f'], w_hh_r=d['w_hh_r'])
      for (o, sv), d in zip(res, types)]
for (a, b) in K.bigru_bwd(bt, bs, T, h):
    outs += [a.cpu(), b.cpu()]
outs += [o.cpu() for o, _ in res]
torch.save(outs, sys.argv[1])
""" % (ROOT,)
    import tempfile
    got = {}
    for ku in ('1', '2'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            r = subprocess.run([sys.executable, '-c', code, f.name], env=dict(os.environ, TWOG_X3S_KU=ku, TWOG_GEMM_XSPLIT='1', TWOG_X3_XL='1', TWOG_X3_ROWS128='0'),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
            got[ku] = torch.load(f.name)
    assert len(got['1']) == len(got['2']) >= 14
    for a, b in zip(got['1'], got['2']):
        assert torch.equal(a, b), float((a - b).abs().max())


def test_gemm_x3_chain_reduction_split_over_workgroups():
    """XL (csrc/gemm_f32.hip::pick_xl_split): the X3 64x64 chain launches of real batches (96+ tiles) split every reduction
    into S slices of 4-wave workgroups and combine them inside the launch by the tile's last arriver, in slice order. In a
    child process per setting (TWOG_X3_XL: 1 = never, 0 = the library's rule, 2 / 3 / 4 forced): plain launches (accumulate,
    bias + ReLU, ragged edges, a grouped launch mixing reduction lengths) against fp64, the BiGRU backward chain (gate
    backward fused into the combining workgroup's epilogue) against the unsplit kernels, every launch twice (bit-identical:
    the slices are added in slice order whoever arrives last) and the tickets back at zero."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels(); DEV = 'cuda:0'; force = sys.argv[2]
g = torch.Generator().manual_seed(12)
outs, split_seen = [], 0
def prob(M, N, Kk, bkm, bias, act, acc):
    A = torch.randn(M, Kk, generator=g).to(DEV)
    B = (torch.randn((Kk, N) if bkm else (N, Kk), generator=g) * 0.1).to(DEV)
    return dict(A=A, B=B, C0=torch.randn(M, N, generator=g).to(DEV), bias=torch.randn(N, generator=g).to(DEV) if bias else None,
                act=act, accumulate=acc)
def run(probs, bkm):
    global split_seen
    res = []
    for rep in range(2):
        ps = [dict({k: v for k, v in p.items() if k != 'C0'}, C=p['C0'].clone()) for p in probs]
        K.gemm(ps, b_kmajor=bkm, chain=True)
        res.append([p['C'] for p in ps])
    cls = K.gemm_last_class()
    assert cls & K.GEMM_X3, hex(cls)
    assert force in ('0', '1') or cls & K.GEMM_XSPLIT, hex(cls)
    assert force != '1' or not cls & K.GEMM_XSPLIT, hex(cls)
    split_seen += bool(cls & K.GEMM_XSPLIT)
    for p, C, C2 in zip(probs, res[0], res[1]):
        assert torch.equal(C, C2), 'launch-to-launch difference'
        ref = p['A'].double() @ (p['B'].double() if bkm else p['B'].double().t())
        if p['bias'] is not None: ref = ref + p['bias'].double()
        if p['accumulate']: ref = ref + p['C0'].double()
        if p['act']: ref = torch.relu(ref)
        err = (C.double() - ref).abs().max().item()
        assert err <= 2e-6 * ref.abs().max().item(), (err / ref.abs().max().item(), tuple(C.shape))
        outs.append(C.cpu())
run([prob(1408, 512, 1536, True, False, 0, True)], True)     # BiGRU backward carry, bs64: 176 tiles
run([prob(1280, 512, 1536, True, False, 0, True)], True)     # segment backward carry: 160 tiles
run([prob(1280, 1024, 1536, True, False, 0, False)], True)   # segment d_mg: 320 tiles
run([prob(1000, 520, 1024, False, True, 1, False)], False)   # ragged rows and columns, bias + ReLU
run([prob(1040, 512, 1536, True, False, 0, True)], True)     # 136 tiles, the last row tile ragged (beyond the 32-row class: 264 tiles of 32 x 64)
run([prob(640, 1536, 512, False, True, 0, False), prob(640, 1536, 1024, False, False, 0, True)], False)   # mixed K
h, bs, T = 512, 64, 4
types = []
for i, E in enumerate((2, 8, 1)):
    types.append(dict(gi=torch.randn(bs, T, E, 6 * h, generator=g).to(DEV), w_hh_f=(torch.randn(3 * h, h, generator=g) * 0.07).to(DEV),
                      b_hh_f=torch.randn(3 * h, generator=g).to(DEV), w_hh_r=(torch.randn(3 * h, h, generator=g) * 0.07).to(DEV),
                      b_hh_r=torch.randn(3 * h, generator=g).to(DEV)))
res = K.bigru_fwd(types, bs, T, h)
bt = [dict(d_out=torch.randn(o.shape, generator=g).to(DEV), save=sv, out=o, w_hh_f=d['w_hh_f'], w_hh_r=d['w_hh_r'])
      for (o, sv), d in zip(res, types)]
first = [(a.clone(), b.clone()) for a, b in K.bigru_bwd(bt, bs, T, h)]
cls = K.gemm_last_class()
assert cls & K.GEMM_GATE and cls & K.GEMM_X3, hex(cls)
assert force == '0' or bool(cls & K.GEMM_XSPLIT) == (force != '1'), hex(cls)
for (a, b), (a2, b2) in zip(first, K.bigru_bwd(bt, bs, T, h)):
    assert torch.equal(a, a2) and torch.equal(b, b2), 'gate-fused chain: launch-to-launch difference'
    outs += [a.cpu(), b.cpu()]
tickets = K._ws[(str(DEV), 'chain', int(K._stream() or 0))][:4096].view(torch.int32)
assert int(tickets.abs().max()) == 0, 'tickets not returned to zero'
torch.save(dict(outs=outs, split_seen=split_seen), sys.argv[1])
""" % (ROOT,)
    import tempfile
    got = {}
    for force in ('1', '0', '2', '3', '4'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            r = subprocess.run([sys.executable, '-c', code, f.name, force], env=dict(os.environ, TWOG_X3_XL=force, TWOG_GEMM_XSPLIT='1', TWOG_X3_ROWS128='0'),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, force + ': ' + r.stdout[-1500:] + r.stderr[-3000:]
            got[force] = torch.load(f.name)
    assert got['1']['split_seen'] == 0 and got['0']['split_seen'] >= 3 and got['4']['split_seen'] == 6
    base = got['1']['outs']
    for force in ('0', '2', '3', '4'):
        assert len(got[force]['outs']) == len(base) >= 13
        for a, b in zip(base, got[force]['outs']):
            assert torch.isfinite(b).all()
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-7, (force, float((a - b).abs().max()), float(a.abs().max()))


def test_gemm_x3_chain_tiles_of_128_rows():
    """Chain launches of the X3 64x64 class whose rows allow it run 128 x 64 tiles (gemm_x3su128_kernel, rows128_pays): a tile
    takes in (128 + 64) K operand values for twice the products of (64 + 64) K. In a child process per setting
    (TWOG_X3_ROWS128 = 0 / 1): the segment level's backward projection launch at bs64 (four problems with two output widths:
    480 -> 240 tiles), d_mg alone (320 -> 160), the sender MLPs with bias + ReLU (row-major B), a ragged last row tile, and a
    shape the rule leaves alone (176 tiles -> 88: fewer than half the CUs). Against fp64; the two tilings agree to 2e-6 (the
    64-row kernels of the smaller launches split k between two wave groups: another summation order); every launch is
    bit-identical from launch to launch."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import twog_gcn_amd
from twog_gcn_amd.kernels import get_kernels
K = get_kernels(); DEV = 'cuda:0'; on = sys.argv[2] == '1'
g = torch.Generator().manual_seed(14)
outs, taken = [], []
def prob(M, N, Kk, bkm, bias, act, acc):
    A = torch.randn(M, Kk, generator=g).to(DEV)
    B = (torch.randn((Kk, N) if bkm else (N, Kk), generator=g) * 0.1).to(DEV)
    return dict(A=A, B=B, C0=torch.randn(M, N, generator=g).to(DEV), bias=torch.randn(N, generator=g).to(DEV) if bias else None,
                act=act, accumulate=acc)
def run(probs, bkm, expect):
    res = []
    for rep in range(2):
        ps = [dict({k: v for k, v in p.items() if k != 'C0'}, C=p['C0'].clone()) for p in probs]
        K.gemm(ps, b_kmajor=bkm, chain=True)
        res.append([p['C'] for p in ps])
    cls = K.gemm_last_class()
    assert cls & K.GEMM_X3, hex(cls)
    t128 = bool(cls & K.GEMM_WAVES8) and not cls & K.GEMM_KSPLIT and not cls & K.GEMM_TILE128
    assert t128 == (on and expect), (hex(cls), on, expect)
    taken.append(t128)
    for p, C, C2 in zip(probs, res[0], res[1]):
        assert torch.equal(C, C2), 'launch-to-launch difference'
        ref = p['A'].double() @ (p['B'].double() if bkm else p['B'].double().t())
        if p['bias'] is not None: ref = ref + p['bias'].double()
        if p['accumulate']: ref = ref + p['C0'].double()
        if p['act']: ref = torch.relu(ref)
        err = (C.double() - ref).abs().max().item()
        assert err <= 2e-6 * ref.abs().max().item(), (err / ref.abs().max().item(), tuple(C.shape))
        outs.append(C.cpu())
run([prob(256, 512, 1536, True, False, 0, True), prob(256, 1024, 1536, True, False, 0, False),
     prob(1024, 512, 1536, True, False, 0, True), prob(1024, 1024, 1536, True, False, 0, False)], True, True)   # 480 -> 240 tiles
run([prob(1280, 1024, 1536, True, False, 0, False)], True, True)     # 320 -> 160
run([prob(1280, 1024, 512, False, True, 1, False)], False, True)     # sender MLPs: bias + ReLU, row-major B
run([prob(1290, 1024, 1024, True, False, 0, True)], True, True)      # ragged last row tile (11 x 16 = 176 tiles)
run([prob(1408, 512, 1536, True, False, 0, True)], True, False)      # 88 tiles of 128 rows: left to the 64-row kernels
torch.save(dict(outs=outs, taken=taken), sys.argv[1])
""" % (ROOT,)
    import tempfile
    got = {}
    for on in ('0', '1'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            r = subprocess.run([sys.executable, '-c', code, f.name, on], env=dict(os.environ, TWOG_X3_ROWS128=on, TWOG_X3_XL='1', TWOG_GEMM_XSPLIT='1'),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, on + ': ' + r.stdout[-1500:] + r.stderr[-3000:]
            got[on] = torch.load(f.name)
    assert got['0']['taken'] == [False] * 5 and got['1']['taken'] == [True, True, True, True, False]
    for a, b in zip(got['0']['outs'], got['1']['outs']):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-7


def test_ssp_gather_with_segment_level_placement(K):
    """Weights stored [time][clip][natt] and the gradient rows a column block of wider rows (the segment level's layout)."""
    bs, T, H, O, cols = 3, 5, 2, 8, 48
    natt = H * H + 2 * H * O + O * O
    att = torch.softmax(rnd(T, bs, natt, seed=1), -1)
    dgi_full = rnd(bs * T * O, 2 * cols, seed=2)
    off = H * H + H * O
    qc = F.ssp_gather(dgi_full[:, cols:], att, natt, bs * natt, off, bs * T, T, H, O)
    qg = K.ssp_gather(dgi_full.to(DEV)[:, cols:], att.to(DEV), natt, bs * natt, off, bs * T, T, H, O)
    close(qg, qc, what='ssp_gather')
    # against the definition
    w = att.permute(1, 0, 2)[..., off:off + O * H].reshape(bs * T, O, H)
    ref = torch.einsum('nkh,nkc->nhc', w, dgi_full[:, cols:].reshape(bs * T, O, cols)).reshape(bs * T * H, cols)
    close(qc, ref, what='specification vs definition')
