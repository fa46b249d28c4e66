#!/usr/bin/env python3
"""Randomised check of the match stage's dispatch (GPU box): random searched-image sizes and batch sizes through
melf_match_ccoeff at DEFAULT dispatch (tuned kernel in whatever layout the planner picks, general kernel, VALU kernel)
against the VALU kernel's whole map on the same images; reports which kernels / layouts were hit.
    python3 tools/fuzz_match.py [seconds]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from meterelf_amd import MeterReader, _params
from meterelf_amd._engine import load_template

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
params = _params.load(pfile)
tpl = load_template(params)
(th, tw) = tpl.shape
auto = MeterReader(params)
os.environ['MELF_MATCH'] = 'dot4'
ref = MeterReader(params)
del os.environ['MELF_MATCH']
rng = np.random.default_rng(int(time.time()) & 0xffff)
hit = collections.Counter()
(t0, cases, bad, last) = (time.time(), 0, 0, time.time())
while time.time() - t0 < budget:
    rows = int(rng.integers(th, th + 150))
    cols = int(rng.choice([tw, tw + 1, tw + 31, tw + 32, tw + 33, tw + 62, tw + 63, int(rng.integers(tw, tw + 140))]))
    n = int(rng.choice([1, 31, 33, 200, 289, 320, 481, 512, 609, 700, 737, 850, 993, 1024, 1089, 1100, int(rng.integers(1, 1400))]))
    n = max(1, min(n, int(6e7 // (rows * cols))))
    imgs = rng.integers(0, 256, size=(n, rows, cols), dtype=np.uint8)
    k = int(rng.integers(0, n))
    if rows > th + 3 and cols > tw + 2:
        imgs[k, 3:3 + th, 2:2 + tw] = tpl
    imgs[-1] = int(rng.integers(0, 256))
    (mv, mx, my, rmap) = auto.ctx.match_ccoeff(imgs, want_map=True)
    info = auto.ctx.last_match()
    hit[(info['kernel'], info['layout'])] += 1
    (mvd, mxd, myd, rmapd) = ref.ctx.match_ccoeff(imgs, want_map=True)
    cases += 1
    if not (np.array_equal(rmap.view(np.uint32), rmapd.view(np.uint32)) and mv.tobytes() == mvd.tobytes() and mx.tobytes() == mxd.tobytes()
            and my.tobytes() == myd.tobytes()):
        bad += 1
        print('MISMATCH rows %d cols %d n %d: %s' % (rows, cols, n, info), flush=True)
    if time.time() - last > 60:
        last = time.time()
        print('... %d cases, %d mismatches' % (cases, bad), flush=True)
print('fuzz: %d cases, %d mismatches, %.0f s; kernels / layouts hit: %s' % (cases, bad, time.time() - t0, dict(hit)))
sys.exit(1 if bad else 0)
