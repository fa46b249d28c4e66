#!/usr/bin/env python3
"""End-to-end files/s of get_meter_values over the fixture files: JPEG decode on the host
(Pillow, 1 and 8 threads) against JPEG decode on the GPU, plus the decode kernels' own timings."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from meterelf_amd import MeterReader, _hip, _params, get_meter_values

d = os.path.join(ROOT, 'tests', 'golden', 'sample-images2')
pfile = os.path.join(d, 'params.yml')
files = sorted(glob.glob(os.path.join(d, '*.jpg'))) * 32


def run(label):
    # warm-up with the same chunk size: the first chunk of a size in a process pays for first-time device and pinned
    # allocations of that size (hundreds of milliseconds), later readers get them back at once
    list(get_meter_values(pfile, files[:2 * int(os.environ.get('METERELF_BATCH', '64'))]))
    t0 = time.perf_counter()
    vals = [r.value for r in get_meter_values(pfile, files)]
    dt = time.perf_counter() - t0
    print('%-28s %d files, %.0f files/s' % (label, len(files), len(files) / dt))
    return vals


os.environ['METERELF_DECODE'] = 'host'
ref = None
for threads in (1, 8):
    os.environ['METERELF_DECODE_THREADS'] = str(threads)
    ref = run('host decode, %d thread(s)' % threads)
os.environ['METERELF_DECODE'] = 'gpu'
for batch in (64, 256, 1024):
    os.environ['METERELF_BATCH'] = str(batch)
    got = run('GPU decode, chunks of %d' % batch)
    assert got == ref, 'values differ between decode paths'
# a list long enough for the per-call costs (context creation and release, ~30 ms) to stop mattering
short = files
files = files * 4
got = run('GPU decode, chunks of 1024')
assert got == ref * 4
for batch in (2048, 4096):
    os.environ['METERELF_BATCH'] = str(batch)
    got = run('GPU decode, chunks of %d' % batch)
    assert got == ref * 4
os.environ['METERELF_BATCH'] = '1024'
files = short

# the decode + read call alone, file bytes already in memory
reader = MeterReader(_params.load(pfile))
blobs = [open(f, 'rb').read() for f in files]
(H, W, ok, _) = _hip.jpeg_probe(blobs[0])
for n in (256, 1024, 4096):
    reader.ctx.jpeg_process_batch(blobs[:n], H, W)
    reader.ctx.jpeg_process_batch(blobs[:n], H, W)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        (recs, status) = reader.ctx.jpeg_process_batch(blobs[:n], H, W)
    dt = (time.perf_counter() - t0) / reps     # no event records: they keep the chunks' kernels from overlapping
    print('melf_jpeg_process_batch n=%d: %.2f ms/call = %.0f files/s' % (n, dt * 1e3, n / dt))
    assert (status == 0).all()
reader.close()
