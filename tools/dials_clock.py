#!/usr/bin/env python3
"""Phase times of k_dials waves from in-kernel stamps (diagnostic build `make -C meterelf_amd/csrc stamp`)."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_stamp.so'))
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

sd = sys.argv[1] if len(sys.argv) > 1 else 'sample-images1'
pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
B = 1024
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), B, 2024, dev)
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    ctx.process_batch_dev(frames.data_ptr(), B, H, W, want_host=False, stream=stream)
torch.cuda.synchronize()
n = 4096
buf = np.zeros((n, 8), np.uint64)
assert _hip.lib().melf_debug_dials_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf[:, :6].astype(np.float64)
names = ['partials min/max + sync', 'dial colour (5x5 mean)', 'window pixels -> in-range mask', 'closing, flood, labelling, areas', 'momentum + angle passes']
tot = t[:, 5] - t[:, 0]
print('waves %d, total cycles per wave: median %.0f p90 %.0f' % (n, np.median(tot), np.sort(tot)[n * 9 // 10]))
t8 = buf.astype(np.float64)
for (label, a, b) in (('  pixel phase: candidate prefilter', 2, 6), ('  pixel phase: scan + list', 6, 7), ('  pixel phase: exact test of candidates', 7, 3)):
    dlt = t8[:, b] - t8[:, a]
    print('  %-42s median %8.0f cycles  p90 %8.0f' % (label, np.median(dlt), np.sort(dlt)[n * 9 // 10]))
for k in range(5):
    dlt = t[:, k + 1] - t[:, k]
    print('  %-34s median %8.0f cycles (%4.1f %%)  p90 %8.0f' % (names[k], np.median(dlt), 100 * np.median(dlt) / np.median(tot), np.sort(dlt)[n * 9 // 10]))
# per dial (wave w of a workgroup = dial w): which dials are the slow ones
nd = int(ctx.params.ndials)
for d in range(nd):
    sel = np.arange(n) % nd == d
    ex = t8[sel, 3] - t8[sel, 7]
    pf = t8[sel, 6] - t8[sel, 2]
    print('  dial %d: total median %6.0f p90 %6.0f | prefilter median %6.0f | exact test median %6.0f p90 %6.0f max %6.0f' % (
        d, np.median(tot[sel]), np.sort(tot[sel])[sel.sum() * 9 // 10], np.median(pf), np.median(ex), np.sort(ex)[sel.sum() * 9 // 10], ex.max()))
