#!/usr/bin/env python3
"""What do the k_match_mfma waves lose to a kernel that is RESIDENT beside them, by the KIND of work it does?  (round 6)
Needs a library whose large match tiles do not name their high register (`make -C meterelf_amd/csrc open` -> csrc/libmeterelf_hip_open.so)
and tools/ubench/libpartner.so (hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/partner.hip -o tools/ubench/libpartner.so).  Context steps on stream 1 with stamps on k_match; on stream 2 a stream of short
partner workgroups of one kind (tools/ubench/partner.hip).  Prints k_match's launch time and the step time per partner.
    python3 tools/corun_partner.py"""
import ctypes as C
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_open.so'))
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

P = C.CDLL(os.path.join(ROOT, 'tools', 'ubench', 'libpartner.so'))
P.partner_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
A = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
base = np.stack([imread_bgr(f) for f in files if imread_bgr(f).shape == imread_bgr(files[0]).shape])
(B, NB) = (1024, 4)
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), NB * B, 2024, dev)
(H, W) = base.shape[1:3]
recs = torch.empty((NB, B * _hip.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
s1 = torch.cuda.Stream(device=dev)
s2 = torch.cuda.Stream(device=dev)
NAMES = {0: 'vector ALU only', 1: 'loads, 256 KiB buffer (L2)', 2: 'loads, streaming 1 GiB', 3: 'loads, streaming, non-temporal', 4: 'stores, streaming',
         5: 'stores, streaming, non-temporal'}


def run(mode, gap, iters=60):
    A.set_profiling(2)
    A.timings()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        A.process_batch_dev(frames.data_ptr() + (i % NB) * B * H * W * 3, B, H, W, want_host=False, stream=s1.cuda_stream, d_results_ptr=recs[i % NB].data_ptr())
        if mode is not None:   # ~one partner launch per step: 16 384 short workgroups (64 iterations each)
            assert P.partner_launch(mode, 16384, 64, gap, big.data_ptr(), big.numel(), s2.cuda_stream) == 0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    (ms, n) = A.timings()['k_match']
    return ms / max(n, 1), dt / iters * 1e3


run(None, 0, 30)
print('k_match %.4f ms, step %.4f ms  -- alone' % run(None, 0))
for mode in (0, 1, 2, 3, 4, 5):
    for gap in (0, 4):
        # the partner alone: how long one launch of it takes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            P.partner_launch(mode, 16384, 64, gap, big.data_ptr(), big.numel(), s2.cuda_stream)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t0) / 10 * 1e3
        (m, step) = run(mode, gap)
        print('k_match %.4f ms, step %.4f ms  -- beside: %s, gap %d (a partner launch alone: %.3f ms)' % (m, step, NAMES[mode], gap, alone))
print('k_match %.4f ms, step %.4f ms  -- alone' % run(None, 0))
