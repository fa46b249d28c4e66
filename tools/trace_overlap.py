#!/usr/bin/env python3
"""Time line of a rocprofv3 --kernel-trace run of the full path: per kernel name the average duration, and for consecutive
steps how much of each k_prep_lplane / k_dials launch lies inside a k_match launch of ANOTHER step (the two-lane overlap).
    python3 tools/trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if 'melf' in r['Kernel_Name']]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void melf::', '')[:28]) for r in rows))
t0 = ev[0][0]
skip = len(ev) // 3          # warm-up third
body = ev[skip:]
match = [(a, b) for (a, b, n) in body if n.startswith('k_match')]
span = (body[-1][1] - body[0][0]) / 1e3
print('%d launches, %d match launches over %.1f us: %.2f us per step' % (len(body), len(match), span, span / max(len(match), 1)))
for kind in ('k_prep_lplane', 'k_match', 'k_dials'):
    sel = [(a, b) for (a, b, n) in body if n.startswith(kind)]
    if not sel:
        continue
    dur = sum(b - a for (a, b) in sel) / len(sel) / 1e3
    inside = 0.0
    if not kind.startswith('k_match'):
        for (a, b) in sel:
            inside += sum(max(0, min(b, mb) - max(a, ma)) for (ma, mb) in match)
        inside /= sum(b - a for (a, b) in sel)
    print('  %-16s n %4d  avg %7.1f us   share of its time inside a k_match launch: %.2f' % (kind, len(sel), dur, inside))
print('first 24 launches of the measured part (us from its start):')
for (a, b, n) in body[:24]:
    print('  %9.1f .. %9.1f  %6.1f  %s' % ((a - body[0][0]) / 1e3, (b - body[0][0]) / 1e3, (b - a) / 1e3, n))
