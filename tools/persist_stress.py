#!/usr/bin/env python3
"""Root-cause rig for the persistent segment launches: runs twog_segrnn_fwd_persistent
of the library TWOG_LIB_PATH points at (the shipped one, `make jitter`, `make diag`: the round-5 P2 variant that needs scratch
memory) against the launch-per-step path of the same library and says WHERE they differ: per buffer the number of wrong
words and the first / last wrong (direction, time step, clip, entity, column), per clip chunk, and whether two runs of the
persistent launch differ from each other (a race) or agree (arithmetic / code generation).
usage: TWOG_LIB_PATH=... python3 tools/persist_stress.py bs T H O h [repeats] [max_chunks]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bs, T, H, O, h = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
if len(sys.argv) > 7:
    os.environ['TWOG_SP_MAX_CHUNKS'] = sys.argv[7]
os.environ['TWOG_PERSIST_CHECK'] = 'sync'
import torch  # noqa: E402
import twog_gcn_amd  # noqa: E402,F401
from twog_gcn_amd import kernels  # noqa: E402
from tests.test_kernels_gpu import _seg_params  # noqa: E402

K = kernels.get_kernels()
DEV = 'cuda:0'
pg = _seg_params(DEV, bs, T, H, O, h, (True, True, True, True), True)
os.environ['TWOG_SEG_PERSIST'] = '0'
ref = {k: v.clone() for k, v in K.segrnn_fwd(pg).items() if torch.is_tensor(v)}
assert not K.last_segrnn_persistent
del os.environ['TWOG_SEG_PERSIST']
keys = ['hs_h', 'hs_o', 'mg_h', 'mg_o', 'msrc_h', 'msrc_o', 'save_h', 'save_o', 'att']
n_chunks = min(int(os.environ.get('TWOG_SP_MAX_CHUNKS', 16)), min(16, 256 // (8 * (h // 16))), bs)
cpc = -(-bs // n_chunks)
print(f'lib {os.environ.get("TWOG_LIB_PATH", "default")} shape bs {bs} T {T} H {H} O {O} h {h}: {-(-bs // cpc)} chunks of {cpc} clips, '
      f'{2 * -(-bs // cpc) * 4 * (h // 16)} workgroups', flush=True)
prev = None
for rep in range(reps):
    got = K.segrnn_fwd(pg)
    assert K.last_segrnn_persistent, 'the persistent launch did not run'
    torch.cuda.synchronize()
    got = {k: got[k].clone() for k in keys}
    line = []
    for k in keys:
        a, b = got[k], ref[k]
        bad = ~torch.isclose(a, b, rtol=5e-5, atol=5e-6)
        if k in ('hs_h', 'hs_o'):       # [bs][T][E][2h]
            view = bad.view(bs, T, -1, 2, h).permute(3, 1, 0, 2, 4)           # dir, t, b, e, col
        elif k == 'att':                # [2][T][bs][natt]
            view = bad.view(2, T, bs, 1, -1)
        else:                           # [2][bs][T][E][w]
            view = bad.view(2, bs, T, bad.shape[3], -1).permute(0, 2, 1, 3, 4)
        n = int(bad.sum())
        if n:
            idx = view.nonzero()
            # earliest wrong step per direction: direction 0 counts t up, direction 1 down
            first = {}
            for d in range(2):
                sel = idx[idx[:, 0] == d]
                if len(sel):
                    tt = sel[:, 1]
                    s_first = int(tt.min()) if d == 0 else int(T - 1 - tt.max())
                    at = sel[(tt == (s_first if d == 0 else T - 1 - s_first))]
                    first[d] = dict(step=s_first, clips=sorted(set(at[:, 2].tolist()))[:8], ents=sorted(set(at[:, 3].tolist()))[:12],
                                    cols=(int(at[:, 4].min()), int(at[:, 4].max())), words=len(at))
            if k == 'hs_o' and os.environ.get('TWOG_STRESS_DETAIL') and 0 in first:
                # the first wrong step of direction 0 in detail: per (clip, entity) row the number of wrong columns
                s0 = first[0]['step']
                at = idx[(idx[:, 0] == 0) & (idx[:, 1] == s0)]
                rows = {}
                for _, _, b_, e_, c_ in at.tolist():
                    rows.setdefault((b_, e_), []).append(c_)
                print('   detail hs_o dir 0 step', s0, {r: (len(c), min(c), max(c)) for r, c in sorted(rows.items())[:40]}, flush=True)
            chunks = sorted(set((idx[:, 2] // cpc).tolist()))
            line.append(f'  {k}: {n} wrong words, chunks {chunks}, first wrong step per direction {first}, '
                        f'max |diff| {float((a - b).abs().max()):.3e}')
    same = None if prev is None else all(torch.equal(prev[k], got[k]) for k in keys)
    print(f'run {rep}: {"OK" if not line else "WRONG"}; identical to the previous run: {same}', flush=True)
    for l in line:
        print(l, flush=True)
    prev = got
