#!/usr/bin/env python3
"""Time budget of ONE bench step from a rocprofv3 --kernel-trace csv (steps are delimited by the fused Adam kernel).
usage: python3 tools/step_breakdown.py <rocprof output dir>"""
import collections
import csv
import glob
import re
import sys

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = adam[-2] + 1, adam[-1] + 1   # the last complete step
step = rows[lo:hi]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])


def cls(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    if 'gemm_gate_bwd_xs' in n:
        return 'recurrence: GEMM NT + fused gate backward, reduction split over workgroups'
    if 'gemm_gate_bwd_x3s' in n:
        return 'recurrence: 64x64 GEMM NT + fused gate backward (bf16 x 3)'
    if 'gemm_gate_bwd' in n:
        return 'recurrence: 64x64 GEMM NT + fused gate backward'
    if 'gemm_x3s_kernel' in n:
        return 'recurrence: 64x64 GEMM (bf16 x 3) ' + ('8 waves, k-split' if ', 2>' in n else '4 waves') + (' NT' if '<true' in n else ' NN')
    if 'gemm_x3_kernel' in n:
        return 'big GEMM 128x128 (bf16 x 3) ' + ('NN (forward)' if '<false, false' in n else 'NT (dX)' if '<false, true' in n else 'TT (dW)' if '<true, true' in n else 'TN')
    if 'gemm_gru_fwd' in n:
        return 'recurrence: frame-level GRU step (W_hh product + gates, one launch)'
    if 'gemm_xs' in n:
        return 'recurrence: GEMM with the reduction split over workgroups (in-launch combine)'
    if 'gemm_ks_kernel' in n or 'gemm_ks32' in n:
        return 'recurrence: 64x64 / 32x64 GEMM (k-split in the workgroup)'
    if 'gemm_kernel<64' in n:
        return 'recurrence: 64x64 GEMM ' + ('NN' if 'false, false' in n else 'NT' if 'false, true' in n else 'TT/TN')
    if 'gemm_kernel<128' in n:
        return 'big GEMM 128x128 ' + ('NN (forward)' if 'false, false' in n else 'NT (dX)' if 'false, true' in n else 'TT (dW)' if 'true, true' in n else 'TN')
    if 'gru_step' in n:
        return 'recurrence: gate kernels'
    if 'attn_' in n and 'gcn' not in n:
        return ('recurrence: segment attention' if g <= 1024 else 'frame-level attention') + (' bwd' if 'bwd' in n else ' fwd')
    m = re.match(r'(void )?([A-Za-z0-9_:]+)', n)
    return 'other: ' + (m.group(2) if m else n)[:48]


agg = collections.defaultdict(lambda: [0.0, 0])
busy = 0.0
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    a = agg[cls(r)]
    a[0] += d
    a[1] += 1
    busy += d
wall = (t1 - t0) / 1e6
print(f'step wall {wall:.2f} ms, sum of kernel durations {busy:.2f} ms, {len(step)} dispatches')
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if ms < 0.05:
        continue
    print(f'{k:52s} {ms:7.2f} ms {100 * ms / wall:5.1f}%  {n:5d} launches  avg {1e3 * ms / n:7.1f} us')
