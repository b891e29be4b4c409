#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats csv pair: per-kernel ms/step and per-call distribution."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
nst = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
stats = glob.glob(d + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls']) / nst:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / nst:7.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:8.1f} pct={float(r['Percentage']):5.1f}")
print('total kernel ms/step', tot / 1e6 / nst)
trace = glob.glob(d + '/*/*_kernel_trace.csv')[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    n = r['Kernel_Name']
    if any(k in n for k in ('attn_', 'gru_step', 'wcolsum')):
        per[n[:50]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in per.items():
    v2 = sorted(v)
    print(f'{k:50s} n={len(v):5d} min {v2[0]:7.1f} med {v2[len(v2) // 2]:7.1f} max {v2[-1]:8.1f} us')
