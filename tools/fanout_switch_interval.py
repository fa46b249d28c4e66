#!/usr/bin/env python3
"""get_meter_values over two contexts on one GPU (METERELF_DEVICES=0,0) against the interpreter's thread switch interval:
the fan-out's worker threads and the consumer hand the GIL to each other once per chunk, and a thread that wants it while
another runs Python code waits up to one interval (5 ms by default -- several chunks' worth).
    python3 tools/fanout_switch_interval.py"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meterelf_amd import get_meter_values, release_cached_contexts

d = os.path.join(ROOT, 'tests', 'golden', 'sample-images1')
pfile = os.path.join(d, 'params.yml')
files = [f for f in sorted(glob.glob(os.path.join(d, '*.jpg')))][2:]
names = [files[i % len(files)] for i in range(64 * 1024)]
for devs in ('0', '0,0', '0,0,0'):
    os.environ['METERELF_DEVICES'] = devs
    for si in (0.005, 0.001, 0.0002):
        os.environ['METERELF_SWITCH_INTERVAL'] = str(si)
        sys.setswitchinterval(0.005)
        sum(1 for _ in get_meter_values(pfile, names[:4096]))
        t0 = time.perf_counter()
        n = sum(1 for r in get_meter_values(pfile, names) if r.error is None)
        dt = time.perf_counter() - t0
        print('METERELF_DEVICES=%-6s switch interval %.4f s: %d files, %.0f files/s' % (devs, si, n, len(names) / dt), flush=True)
    release_cached_contexts()
