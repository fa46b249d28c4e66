#!/usr/bin/env python3
"""What does k_match_mfma lose when another kernel shares the chip?  Context A runs the full path (1024 frames, stream 1, dispatch
stamps on k_match); context B meanwhile runs the fused full-frame mask kernel (<= 64 registers: resident beside the match waves;
coalesced 16-byte loads, HBM-bound) on stream 2.  Prints k_match's average launch time alone and with the partner.
    python3 tools/corun_probe.py"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
A = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
Bc = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
base = np.stack([imread_bgr(f) for f in files if imread_bgr(f).shape == imread_bgr(files[0]).shape])
B = 1024
NB = 4
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), NB * B, 2024, dev)
(H, W) = base.shape[1:3]
recs = torch.empty((NB, B * _hip.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
FB = 256
fr2 = torch.randint(0, 256, (4 * FB, 480, 640, 3), dtype=torch.uint8, device=dev, generator=g)
mk2 = torch.empty((4 * FB, 480, 640), dtype=torch.uint8, device=dev)
s1 = torch.cuda.Stream(device=dev)
s2 = torch.cuda.Stream(device=dev)


def run(partner, iters=120):
    A.set_profiling(2)
    A.timings()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        A.process_batch_dev(frames.data_ptr() + (i % NB) * B * H * W * 3, B, H, W, want_host=False, stream=s1.cuda_stream, d_results_ptr=recs[i % NB].data_ptr())
        if partner:
            for j in range(3):   # ~3 x 65 us of streaming per 270 us step
                b = (3 * i + j) % 4
                Bc.hls_inrange_close_dev(fr2.data_ptr() + b * FB * 480 * 640 * 3, FB, 480, 640, mk2.data_ptr() + b * FB * 480 * 640, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    (ms, n) = A.timings()['k_match']
    return ms / max(n, 1), dt / iters * 1e3


run(False, 40)
for (label, p) in (('alone', False), ('with the fused mask kernel streaming on another stream', True), ('alone', False)):
    (m, step) = run(p)
    print('k_match %.4f ms per launch, %.4f ms per step  -- %s' % (m, step, label))
