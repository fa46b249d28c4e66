#!/usr/bin/env python3
"""Randomised JPEG decode soak on the GPU box: random sizes, contents, qualities, sampling modes, Huffman
optimisation and restart intervals, every decoded byte against libjpeg-turbo (Pillow).
    python3 tools/soak_jpeg.py [seconds]"""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from PIL import Image

from meterelf_amd import MeterReader, _params
import test_jpeg as TJ

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
reader = MeterReader(_params.load(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')))
ctx = reader.ctx
rng = np.random.default_rng(20261003)
t0 = time.time()
(nfiles, bad, nbytes) = (0, 0, 0)
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60:  # a sign of life for long runs
        last_note = time.time()
        print('... %d files, %d mismatches, %.0f s' % (nfiles, bad, last_note - t0), flush=True)
    (H, W) = (int(rng.integers(1, 300)), int(rng.integers(1, 400)))
    files = []
    for _ in range(int(rng.integers(1, 24))):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            img = TJ._natural_image(rng, H, W)
        elif kind == 1:
            img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == 2:
            img = np.full((H, W, 3), rng.integers(0, 256, 3), np.uint8)
        else:
            img = np.clip(TJ._natural_image(rng, H, W).astype(np.int16) * 3 - 256, 0, 255).astype(np.uint8)  # saturating
        kw = dict(quality=int(rng.integers(1, 101)), optimize=bool(rng.integers(0, 2)))
        grey = rng.integers(0, 6) == 0
        if not grey:
            kw['subsampling'] = ['4:4:4', '4:2:2', '4:2:0'][int(rng.integers(0, 3))]
        if rng.integers(0, 4) == 0:
            kw['restart_marker_blocks'] = int(rng.integers(1, 9))
        try:
            files.append(TJ._encode(img[..., 0] if grey else img, **kw))
        except OSError:
            pass  # Pillow's encoder refuses some parameter combinations
    if not files:
        continue
    (frames, status) = ctx.jpeg_decode(files, H, W)
    for (i, d) in enumerate(files):
        nfiles += 1
        nbytes += len(d)
        if status[i] != 0 or not np.array_equal(frames[i], TJ._pillow_bgr(d)):
            bad += 1
            print('MISMATCH %dx%d file %d status %d' % (H, W, i, status[i]))
reader.close()
print('jpeg soak: %d files (%.1f MB) decoded, %d mismatches, %.0f s' % (nfiles, nbytes / 1e6, bad, time.time() - t0))
sys.exit(1 if bad else 0)
