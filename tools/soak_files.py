#!/usr/bin/env python3
"""Randomised soak of the file-name API on the GPU box: get_meter_values over random sub-lists of the fixture files with
random chunk sizes (two chunks in flight in the library), consumers that stop half way, files for the host branch and
missing files mixed in; every value against the one-file-at-a-time reading of the same file.
    python3 tools/soak_files.py [seconds]"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image

from meterelf_amd import get_meter_values

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261004)
sets = {}
tmp = '/tmp/melf_soak_files'
os.makedirs(tmp, exist_ok=True)
for sd in ('sample-images1', 'sample-images2'):
    d = os.path.join(ROOT, 'tests', 'golden', sd)
    files = sorted(glob.glob(os.path.join(d, '*.jpg')))
    prog = os.path.join(tmp, sd + '-progressive.jpg')
    Image.open(files[3]).save(prog, 'JPEG', progressive=True, quality=92)   # host branch
    files = files + [prog, os.path.join(tmp, 'missing.jpg')]
    pfile = os.path.join(d, 'params.yml')
    os.environ['METERELF_BATCH'] = '1'
    ref = {r.filename: (r.value, None if r.error is None else r.error.get_message()) for r in get_meter_values(pfile, files)}
    sets[sd] = (pfile, files, ref)
t0 = time.time()
(calls, nfiles, bad, closed) = (0, 0, 0, 0)
last = t0
while time.time() - t0 < budget:
    if time.time() - last > 60:
        last = time.time()
        print('... %d calls, %d files, %d mismatches' % (calls, nfiles, bad), flush=True)
    sd = ('sample-images1', 'sample-images2')[int(rng.integers(0, 2))]
    (pfile, files, ref) = sets[sd]
    n = int(rng.integers(1, 3000))
    lst = [files[i] for i in rng.integers(0, len(files), n)]
    os.environ['METERELF_BATCH'] = str(int(rng.choice([1, 7, 64, 200, 512, 1024])))
    stop = int(rng.integers(0, n)) if rng.random() < 0.3 else n
    gen = get_meter_values(pfile, lst)
    got = []
    for r in gen:
        got.append(r)
        if len(got) >= stop:
            break
    if stop < n:
        gen.close()
        closed += 1
    for (r, f) in zip(got, lst):
        want = ref[f]
        have = (r.value, None if r.error is None else r.error.get_message())
        if r.filename != f or have != want:
            bad += 1
            if bad < 5:
                print('MISMATCH', f, have, want)
    calls += 1
    nfiles += len(got)
print('files soak: %d calls (%d closed half way), %d files, %d mismatches, %.0f s' % (calls, closed, nfiles, bad, time.time() - t0))
sys.exit(1 if bad else 0)
