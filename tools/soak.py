#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: many synthetic batches (shift up to 24 px, noise sigma up to 12, so that
thresholds, failed dials and unreadable needles all occur) through the HIP path and the CPU oracle (checker only);
prints the number of frames compared and of mismatches.     python3 tools/soak.py [seconds] [sample dir ...]"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

from meterelf_amd import MeterReader, _params
from oracle import pyoracle as po
import test_gpu_parity as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dirs = sys.argv[2:] or ['sample-images1', 'sample-images2']
t0 = time.time()
total = bad = 0
stats = {}
seed = 1000
last_note = t0
while time.time() - t0 < budget:
    for sd in dirs:
        if time.time() - last_note > 60:  # a sign of life for long runs
            last_note = time.time()
            print('... %d frames, %d mismatches, %.0f s' % (total, bad, last_note - t0), flush=True)
        pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
        files = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg')))
        if seed % 4:   # mostly the readable frames; every fourth round the two rejected (other orientation) ones of sample-images1
            files = T._good(files)
        reader = MeterReader(_params.load(pfile))
        op = po.Params(pfile)
        seed += 1
        rng = np.random.default_rng(seed)
        nfr = int(rng.choice([1, 31, 64, 97, 200]))   # batch sizes on both sides of the kernels' dispatch thresholds
        frames = T.synth_frames(files, nfr, seed, shift=int(rng.integers(0, 25)), sigma=float(rng.uniform(0, 16)))
        recs = reader.read_frames(frames)
        ores = po.process_frames(frames, op)
        for i in range(len(frames)):
            total += 1
            stats[ores[i].status] = stats.get(ores[i].status, 0) + 1
            try:
                T._compare_records(recs[i:i + 1], [ores[i]], tag='%s seed %d frame %d' % (sd, seed, i))
            except AssertionError as e:
                bad += 1
                print('MISMATCH', e)
        reader.close()
print('soak: %d frames compared, %d mismatches, oracle statuses %s, %.0f s' % (total, bad, stats, time.time() - t0))
sys.exit(1 if bad else 0)
