#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 --stats kernel_stats.csv:  python3 tools/kstats.py <rocprof dir> [rows]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:84]
    print('%-86s calls %6s avg %9.1f us total %9.2f ms' % (name, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
