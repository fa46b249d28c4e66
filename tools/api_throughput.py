#!/usr/bin/env python3
"""End-to-end files/s of get_meter_values (JPEG decode on the host + GPU path) over the fixtures."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meterelf_amd import get_meter_values
d = os.path.join(ROOT, 'tests', 'golden', 'sample-images2')
files = sorted(glob.glob(os.path.join(d, '*.jpg'))) * 4
for threads in (1, 8):
    os.environ['METERELF_DECODE_THREADS'] = str(threads)
    list(get_meter_values(os.path.join(d, 'params.yml'), files[:64]))
    t0 = time.perf_counter()
    n = sum(1 for r in get_meter_values(os.path.join(d, 'params.yml'), files) if r.value)
    dt = time.perf_counter() - t0
    print('decode threads %d: %d files, %.1f files/s' % (threads, n, len(files) / dt))
