#!/usr/bin/env python3
"""Time line of the LAST melf_jpeg_process_batch call in a rocprofv3 --kernel-trace --memory-copy-trace run of
tools/jpeg_call_rate.py: every kernel and copy with start / end (us from the call's first event), so that what the
chip waits for between the chunks of a call can be read off.
    python3 tools/jpeg_time_line.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [gap us that separates calls]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void melf::', '')[:34],
                   'q' + r.get('Queue_Id', '?')))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy %s %s B' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?'))), ''))
ev.sort()
# calls = runs of events separated by idle gaps
calls = [[ev[0]]]
end = ev[0][1]
for e in ev[1:]:
    if e[0] - end > gap * 1e3:
        calls.append([])
    calls[-1].append(e)
    end = max(end, e[1])
calls = [c for c in calls if sum(1 for e in c if e[2].startswith('k_jpeg_huff')) >= 2]
print('%d calls with Huffman launches; spans (us): %s' % (len(calls), ' '.join('%.0f' % ((max(e[1] for e in c) - c[0][0]) / 1e3) for c in calls[-8:])))
for c in calls[-int(os.environ.get('CALLS', '1')):]:
  t0 = c[0][0]
  busy = 0
  last = t0
  for (a, b, n, q) in c:
      busy += max(0, b - max(a, last))
      last = max(last, b)
  print('last call: %d events over %.1f us, something running %.1f us of it' % (len(c), (last - t0) / 1e3, busy / 1e3))
  for (a, b, n, q) in c:
      print('  %8.1f %8.1f  %7.1f  %-4s %s' % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, n))
