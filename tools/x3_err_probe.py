"""Error statistics (max and rms, relative to the largest output) of one GEMM shape against an fp64 product; run once per
TWOG_GEMM_X3 mode. usage: python tools/x3_err_probe.py M N K [chain]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twog_gcn_amd  # noqa
from twog_gcn_amd.kernels import get_kernels
K = get_kernels()
M, N, Kk = (int(x) for x in sys.argv[1:4])
chain = len(sys.argv) > 4
for seed in range(4):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, Kk, generator=g).cuda()
    B = (torch.randn(Kk, N, generator=g) * 0.1).cuda()
    C = torch.zeros(M, N).cuda()
    K.gemm([dict(A=A, B=B, C=C)], b_kmajor=True, chain=chain, split_k_workspace=not chain)
    ref = A.double() @ B.double()
    e = (C.double() - ref)
    print(f"X3={os.environ.get('TWOG_GEMM_X3', '1')} cls {K.gemm_last_class():#x} seed {seed}: max {float(e.abs().max() / ref.abs().max()):.3e} "
          f"rms {float(e.pow(2).mean().sqrt() / ref.abs().max()):.3e} mean {float(e.mean() / ref.abs().max()):.2e}")
