#!/usr/bin/env python3
"""Idle time between consecutive kernels of the full path, from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/run_stage.py full --iters 40
    python3 tools/kernel_gaps.py DIR
Prints, per pair (previous kernel -> next kernel), the median gap between the end of one and the start of the next."""
import csv
import glob
import os
import sys
from collections import defaultdict

f = [p for p in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)][0]
rows = [r for r in csv.DictReader(open(f)) if 'melf' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.split('(')[0].split('::')[-1].split('<')[0]
gaps = defaultdict(list)
durs = defaultdict(list)
for (a, b) in zip(rows, rows[1:]):
    gaps[(short(a['Kernel_Name']), short(b['Kernel_Name']))].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
for r in rows:
    durs[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
med = lambda v: sorted(v)[len(v) // 2]
for (k, v) in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 5:
        print('%-16s -> %-16s n=%4d  median gap %7.2f us  (p10 %.2f, p90 %.2f)' % (k[0], k[1], len(v), med(v) / 1e3, sorted(v)[len(v) // 10] / 1e3, sorted(v)[len(v) * 9 // 10] / 1e3))
for (k, v) in durs.items():
    if len(v) >= 5:
        print('%-16s n=%4d  median duration %7.2f us' % (k, len(v), med(v) / 1e3))
