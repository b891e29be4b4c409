"""Plane-exactness probe of the X3 kernels: one operand holds full 24-bit significands, the other a signed permutation
(one +-2^e per row / column), so every output is ONE input element times a power of two -- any lost bit of the
(h, m, l) planes shows as a nonzero error. usage: python tools/x3_exact_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()


def run(M, N, Kk, bkm, chain, which):
    g = torch.Generator().manual_seed(1)
    full = lambda *s: torch.randn(*s, generator=g) * 2.0 ** torch.randint(-8, 8, s, generator=g).float()
    if which == 'A':      # C[i, j] = A[i, p[j]] * s[j]
        A = full(M, Kk)
        p = torch.randint(0, Kk, (N,), generator=g)
        sgn = (torch.randint(0, 2, (N,), generator=g) * 2 - 1).float() * 2.0 ** torch.randint(-3, 3, (N,), generator=g).float()
        Bkn = torch.zeros(Kk, N)
        Bkn[p, torch.arange(N)] = sgn
        ref = A[:, p] * sgn
    else:                 # C[i, j] = s[i] * B[p[i], j]
        Bkn = full(Kk, N)
        p = torch.randint(0, Kk, (M,), generator=g)
        sgn = (torch.randint(0, 2, (M,), generator=g) * 2 - 1).float() * 2.0 ** torch.randint(-3, 3, (M,), generator=g).float()
        A = torch.zeros(M, Kk)
        A[torch.arange(M), p] = sgn
        ref = Bkn[p, :] * sgn[:, None]
    B = Bkn if bkm else Bkn.t().contiguous()
    C = torch.zeros(M, N).cuda()
    K.gemm([dict(A=A.cuda(), B=B.cuda(), C=C)], b_kmajor=bkm, chain=chain, split_k_workspace=not chain)
    bad = (C.cpu() != ref)
    rel = ((C.cpu() - ref).abs() / ref.abs().clamp_min(1e-30)).max()
    print(f"{M}x{N}x{Kk} bkm={bkm} chain={chain} full={which}: cls {K.gemm_last_class():#x} mismatches {int(bad.sum())} / {bad.numel()} worst rel {float(rel):.2e}")


for which in 'AB':
    for bkm in (True, False):
        run(1408, 512, 1536, bkm, True, which)      # 64x64, 8 waves (k-split in the workgroup)
        run(1280, 1024, 512, bkm, True, which)
        run(4096, 1024, 512, bkm, True, which)      # 64x64, 4 waves
        run(4096, 512, 1536, bkm, False, which)     # 128x128
