#!/usr/bin/env python3
"""In-kernel clock of k_match_mfma (diagnostic build `make -C meterelf_amd/csrc stamp`, loaded through
MELF_LIB_PATH): per-wave shader cycles, wall time and the clock they imply, after a few seconds of
back-to-back launches on the bench workload (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_stamp.so'))
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr
import glob

pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '*.jpg')))
         if os.path.basename(f) not in bench.REJECTED]
base = np.stack([imread_bgr(f) for f in files if imread_bgr(f).shape == imread_bgr(files[0]).shape])
B = 1024
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), B, 2024, dev)
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
t0 = time.time()
while time.time() - t0 < float(sys.argv[1] if len(sys.argv) > 1 else 3.0):
    for _ in range(50):
        ctx.process_batch_dev(frames.data_ptr(), B, H, W, want_host=False, stream=stream)
    torch.cuda.synchronize()
L = _hip.lib()
n = 1024
buf = np.zeros((n, 4), np.uint64)
assert L.melf_debug_match_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
le = np.zeros(n, np.uint64)
assert L.melf_debug_match_loop_end(le.ctypes.data_as(C.c_void_p), n) == 0
loop_cyc = (le - buf[:, 0]).astype(np.float64)
cyc = (buf[:, 1] - buf[:, 0]).astype(np.float64)
rt = (buf[:, 3] - buf[:, 2]).astype(np.float64) / 100e6  # seconds
clk = cyc / rt / 1e9
span = (buf[:, 3].max() - buf[:, 2].min()) / 100e6
print('waves %d | cycles per wave: median %.0f min %.0f max %.0f | wall per wave: median %.1f us max %.1f us | '
      'clock GHz: median %.3f min %.3f max %.3f | first start to last end %.1f us'
      % (n, np.median(cyc), cyc.min(), cyc.max(), np.median(rt) * 1e6, rt.max() * 1e6, np.median(clk), clk.min(), clk.max(), span * 1e6))
nblk = n
ids = np.arange(n)
per, rem, xcd, sub = nblk // 8, nblk % 8, ids & 7, ids >> 3
vid = np.where(xcd < rem, xcd * (per + 1), rem * (per + 1) + (xcd - rem) * per) + sub
(nparts, na) = (32, 24)
rblk = vid % nparts
for (label, sel) in (('4-row waves (8 half-row units)', rblk < na), ('5-row waves (9 units)', rblk >= na)):
    c = np.sort(cyc[sel]); w = np.sort(rt[sel]) * 1e6
    print('  %-32s start->MFMA loops done p50 %.0f cycles, epilogue p50 %.0f cycles' % (label, np.median(loop_cyc[sel]), np.median(cyc[sel] - loop_cyc[sel])))
    print('  %-32s n=%4d cycles p10 %.0f p50 %.0f p90 %.0f max %.0f | wall us p50 %.1f p90 %.1f max %.1f'
          % (label, sel.sum(), c[len(c) // 10], c[len(c) // 2], c[len(c) * 9 // 10], c[-1], w[len(w) // 2], w[len(w) * 9 // 10], w[-1]))
start = (buf[:, 2] - buf[:, 2].min()) / 100.0
print('  wave start offsets us: p50 %.1f p90 %.1f max %.1f' % (np.median(start), np.sort(start)[n * 9 // 10], start.max()))
for mf in (5760, 6480):   # 8 / 9 half-row units x 120 template rows x 6 matrix instructions
    print('  %d MFMAs x 32 cycles = %d busy cycles = %.0f %% of the median wave' % (mf, mf * 32, 100 * mf * 32 / np.median(cyc)))
