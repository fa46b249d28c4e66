#!/usr/bin/env python3
"""K batches through melf_process_batch_dev one after the other against one melf_process_stream_dev call (the same
batches alternating between the two pipeline lanes): ms per batch, and the records must be identical."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
(B, K) = (1024, 20)
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), 2 * B, 2024, dev)  # two different batches
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
rs = _hip.RESULT_DTYPE.itemsize
res_a = torch.zeros(2 * B * rs, dtype=torch.uint8, device=dev)
res_b = torch.zeros(2 * B * rs, dtype=torch.uint8, device=dev)
bs = B * H * W * 3
for _ in range(3):
    for b in range(2):
        ctx.process_batch_dev(frames.data_ptr() + b * bs, B, H, W, d_results_ptr=res_a.data_ptr() + b * B * rs, want_host=False, stream=stream)
    ctx.process_stream_dev(frames.data_ptr(), 2, bs, B, H, W, res_b.data_ptr(), B, stream=stream)
torch.cuda.synchronize()
assert torch.equal(res_a, res_b), 'records differ between the two entry points'
for rep in range(2):
    t0 = time.perf_counter()
    for k in range(K):
        ctx.process_batch_dev(frames.data_ptr() + (k & 1) * bs, B, H, W, d_results_ptr=res_a.data_ptr() + (k & 1) * B * rs, want_host=False, stream=stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ctx.process_stream_dev(frames.data_ptr(), K, 0, B, H, W, res_b.data_ptr(), 0, stream=stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('one call per batch: %.4f ms/batch | one stream call (two lanes): %.4f ms/batch' % ((t1 - t0) / K * 1e3, (t2 - t1) / K * 1e3))
