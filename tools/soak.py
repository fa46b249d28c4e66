#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: many synthetic batches (shift up to 24 px, noise sigma up to 12, so that
thresholds, failed dials and unreadable needles all occur) through the HIP path and the CPU oracle (checker only);
prints the number of frames compared and of mismatches.     python3 tools/soak.py [seconds] [sample dir ...]

    python3 tools/soak.py --resident [seconds]
Resident mode: batches of 512 / 700 / 1024 / 1056 frames uploaded once and read with ONE melf_process_batch_dev call at
default dispatch -- the tuned matrix-core kernel in the layouts production uses (the default mode's host-fed batches of
<= 200 frames land on the general kernel) -- every record against the general kernel's on the same device buffer, and a
random 96 of them against the oracle."""
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

RESIDENT = '--resident' in sys.argv
argv = [a for a in sys.argv[1:] if a != '--resident']
budget = float(argv[0]) if argv else 60.0
dirs = argv[1:] or ['sample-images1', 'sample-images2']


def resident_soak():
    import ctypes as C
    from helpers import hip_runtime
    sd = 'sample-images1'
    pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
    files = T._good(sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))))
    params = _params.load(pfile)
    op = po.Params(pfile)
    tuned = MeterReader(params)
    os.environ['MELF_MATCH'] = 'gen'
    gen = MeterReader(params)
    del os.environ['MELF_MATCH']
    hip = hip_runtime()
    t0 = time.time()
    (total, sampled, bad, seed, layouts, stats) = (0, 0, 0, 5000, {}, {})
    last = t0
    while time.time() - t0 < budget:
        seed += 1
        rng = np.random.default_rng(seed)
        n = int(rng.choice([512, 700, 1024, 1056]))
        base = T.synth_frames(files, 128, seed, shift=int(rng.integers(0, 25)), sigma=float(rng.uniform(0, 16)))
        frames = base[rng.integers(0, 128, n)]
        for i in rng.choice(n, 8, replace=False):      # per-frame variation on top of the repeats
            frames[i] = np.roll(frames[i], int(rng.integers(-5, 6)), axis=1)
        (H, W) = frames.shape[1:3]
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), C.c_size_t(frames.nbytes)) == 0
        try:
            assert hip.hipMemcpy(d, frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
            recs = tuned.ctx.process_batch_dev(d.value, n, H, W)
            info = tuned.ctx.last_match()
            assert info['kernel'] == 'mfma', info
            layouts[info['layout']] = layouts.get(info['layout'], 0) + 1
            grecs = gen.ctx.process_batch_dev(d.value, n, H, W)
            assert gen.ctx.last_match()['kernel'] == 'gen'
        finally:
            hip.hipFree(d)
        total += n
        if recs.tobytes() != grecs.tobytes():
            diff = [i for i in range(n) if recs[i].tobytes() != grecs[i].tobytes()]
            bad += len(diff)
            print('MISMATCH tuned vs general kernel: seed %d n %d frames %s' % (seed, n, diff[:10]))
        pick = rng.choice(n, 96, replace=False)
        ores = po.process_frames(frames[pick], op)
        for (k, i) in enumerate(pick):
            sampled += 1
            stats[ores[k].status] = stats.get(ores[k].status, 0) + 1
            try:
                T._compare_records(recs[i:i + 1], [ores[k]], tag='resident seed %d frame %d' % (seed, i))
            except AssertionError as e:
                bad += 1
                print('MISMATCH', e)
        if time.time() - last > 60:
            last = time.time()
            print('... %d frames (%d against the oracle), %d mismatches, %.0f s' % (total, sampled, bad, last - t0), flush=True)
    print('resident soak: %d frames through the tuned kernel (layouts %s), all compared with the general kernel, %d with the oracle '
          '(statuses %s), %d mismatches, %.0f s' % (total, layouts, sampled, stats, bad, time.time() - t0))
    tuned.close()
    gen.close()
    sys.exit(1 if bad else 0)


if RESIDENT:
    resident_soak()
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
