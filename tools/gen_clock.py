#!/usr/bin/env python3
"""Phase times of k_match_gen waves from in-kernel stamps (diagnostic build `make -C meterelf_amd/csrc stamp`).
    python3 tools/gen_clock.py [sample dir] [batch] [MELF_GEN_SHAPE]"""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_stamp.so'))
os.environ['MELF_MATCH'] = 'gen'
if len(sys.argv) > 3:
    os.environ['MELF_GEN_SHAPE'] = sys.argv[3]
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

sd = sys.argv[1] if len(sys.argv) > 1 else 'sample-images2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), B, 2025, dev)
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    ctx.process_batch_dev(frames.data_ptr(), B, H, W, want_host=False, stream=stream)
torch.cuda.synchronize()
info = ctx.last_match()
n = min(16384, info['waves'])
buf = np.zeros((n, 12), np.uint64)
assert _hip.lib().melf_debug_gen_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
print('%s n=%d: %s, %d waves (%d read back)' % (sd, B, info['layout'], info['waves'], n))
t = buf[:, :8].astype(np.float64)
kind = buf[:, 8].astype(np.int64)
t0 = t[:, 0].min()
names = ['zero LDS + barrier', 'window sums (before the K loop)', 'K loop', 'LDS adds (+ window sums after)', 'wait at the barrier', 'epilogue share',
         'fold maxima + store']
rt = buf[:, 11].astype(np.float64)
print('  kernel span by the 100 MHz clock of the stamped waves: %.1f us (first wave start to last wave start)' % ((rt.max() - rt.min()) / 100.0))
for (k, label) in ((1, 'H-form waves'), (0, 'V-form waves')):
    m = kind == k
    if not m.any():
        continue
    tt = t[m]
    tot = tt[:, 7] - tt[:, 0]
    print('  %s: %d, lifetime median %.0f cycles p90 %.0f max %.0f; start after the first wave: median %.0f, max %.0f; end: max %.0f' % (
        label, m.sum(), np.median(tot), np.percentile(tot, 90), tot.max(), np.median(tt[:, 0] - t0), (tt[:, 0] - t0).max(), (tt[:, 7] - t0).max()))
    for j in range(7):
        d = tt[:, j + 1] - tt[:, j]
        print('    %-36s median %8.0f  p90 %8.0f  max %8.0f' % (names[j], np.median(d), np.percentile(d, 90), d.max()))
