"""GPU box: signed error statistics of the X3 GEMM kernels (fp32 operands split exactly into three bf16 chunks, 6 of the 9
chunk products kept; csrc/gemm_f32.hip) on operands chosen AGAINST the scheme -- the truncating split makes the three
dropped products (m l, l m, l l) carry the sign of a b, so on same-sign operands their sum is a bias, not noise:

  same-sign  : A = relu(randn), B = |randn| * 0.1          (every product >= 0; K up to 61 440: the dW reductions)
  post-ReLU  : A = relu(randn), B = randn * 0.1            (activations x signed weights: the forward launches)
  spread     : every element scaled by 2^e, e uniform in [-20, 20] (a 2^40 exponent spread inside each dot product)

Per case: mean / rms / max of (C - C64) / sum_k |a_k b_k| -- the natural scale of a K-term fp32 dot product's error -- and the
mean of (C - C64) / C64 where all terms share a sign. Run once per library build (TWOG_LIB_PATH selects it; the class bits
say which kernel ran):   python tools/x3_bias_probe.py  [--json]
tools/x3_products_compare.sh runs it for the shipped library (6 products), a TWOG_X3_PRODUCTS=8 build and TWOG_GEMM_X3=0."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd.kernels import get_kernels  # noqa: E402

DEV = 'cuda:0'


def make(kind, shape, g):
    x = torch.randn(shape, generator=g)
    if kind == 'relu':
        return torch.relu(x)
    if kind == 'abs':
        return x.abs() * 0.1
    if kind == 'w':
        return x * 0.1
    if kind == 'spread':
        return x * torch.exp2(torch.randint(-20, 21, shape, generator=g).float())
    raise ValueError(kind)


CASES = [  # name, M, N, K, a_kmajor, b_kmajor, kind of A, kind of B, chain launch
    ('same-sign TT K=61440 (dW)', 512, 1536, 61440, True, True, 'relu', 'abs', False),
    ('same-sign NN K=2048', 8192, 512, 2048, False, False, 'relu', 'abs', False),
    ('same-sign NT K=1536', 8192, 512, 1536, False, True, 'relu', 'abs', False),
    ('post-ReLU NN K=2048', 8192, 512, 2048, False, False, 'relu', 'w', False),
    ('post-ReLU TT K=61440', 512, 1536, 61440, True, True, 'relu', 'w', False),
    ('spread NN K=1024', 4096, 1024, 1024, False, False, 'spread', 'spread', False),
    ('spread TT K=15360', 1536, 512, 15360, True, True, 'spread', 'spread', False),
    ('same-sign chain NT K=1536', 1408, 512, 1536, False, True, 'relu', 'abs', True),
    ('same-sign chain NN K=512', 1280, 1024, 512, False, False, 'relu', 'abs', True),
]


def run():
    K = get_kernels()
    g = torch.Generator().manual_seed(20240)
    rows = []
    for name, M, N, Kk, akm, bkm, ka, kb, chain in CASES:
        A = make(ka, (Kk, M) if akm else (M, Kk), g).to(DEV)
        B = make(kb, (Kk, N) if bkm else (N, Kk), g).to(DEV)
        C = torch.empty(M, N, device=DEV)
        K.gemm([dict(A=A, B=B, C=C, bias=None, act=0, accumulate=False)], a_kmajor=akm, b_kmajor=bkm, chain=chain,
               split_k_workspace=not chain)
        cls = K.gemm_last_class()
        Ad = (A.t() if akm else A).double()
        Bd = (B if bkm else B.t()).double()
        ref = Ad @ Bd
        mag = Ad.abs() @ Bd.abs()
        e = (C.double() - ref) / mag.clamp_min(1e-300)
        row = dict(case=name, x3=bool(cls & K.GEMM_X3), tile128=bool(cls & K.GEMM_TILE128),
                   mean_err_over_sum_abs=float(e.mean()), rms=float(e.pow(2).mean().sqrt()), max=float(e.abs().max()))
        if ka == 'relu' and kb == 'abs':
            row['mean_rel_err_same_sign'] = float(((C.double() - ref) / ref.clamp_min(1e-300)).mean())
        rows.append(row)
        del A, B, C, Ad, Bd, ref, mag, e
    return rows


if __name__ == '__main__':
    rows = run()
    if '--json' in sys.argv:
        print('ROWS ' + json.dumps(rows))
    else:
        lib = os.environ.get('TWOG_LIB_PATH', 'shipped lib2ggcn_hip.so')
        print(f'# library: {lib}   TWOG_GEMM_X3={os.environ.get("TWOG_GEMM_X3", "1")}')
        print(f'{"case":34s} {"kernel":>14s} {"mean/sum|ab|":>13s} {"rms":>10s} {"max":>10s} {"mean rel (same sign)":>21s}')
        for r in rows:
            kern = ('x3 ' if r['x3'] else 'fp32 ') + ('128' if r['tile128'] else '64')
            ms = f"{r['mean_rel_err_same_sign']:+.2e}" if 'mean_rel_err_same_sign' in r else ''
            print(f"{r['case']:34s} {kern:>14s} {r['mean_err_over_sum_abs']:+13.2e} {r['rms']:10.2e} {r['max']:10.2e} {ms:>21s}")
