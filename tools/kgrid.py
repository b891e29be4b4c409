#!/usr/bin/env python3
"""Launches of the last bench step grouped by (kernel, workgroups): python3 tools/kgrid.py <rocprof dir> [filter]"""
import collections, csv, glob, re, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[adam[-2] + 1:adam[-1] + 1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    if flt not in n:
        continue
    wg = (int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']))
    a = agg[(n[:60], wg)]
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for (n, wg), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%-60s wgs %-12s %4d launches avg %8.1f us total %8.2f ms' % (n, wg, c, us / c, us / 1e3))
