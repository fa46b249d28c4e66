#!/usr/bin/env python3
"""Is k_match_mfma's clock (the launch is power-limited) a matter of WHAT the operands are?  1024 frames, the tuned kernel, event
times of k_match for combinations of template (the real one / flat 128, i.e. A operands all zero / random) and frames (the bench's
synthetic frames / all 128, i.e. B operands all zero / all 0 / random bytes).  Not a parity run: the records are not looked at.
    python3 tools/match_operand_power.py"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

sd = 'sample-images1'
params = _params.load(os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml'))
tpl = _engine.load_template(params)
blob_real = _engine.make_blob(params)
rng = np.random.default_rng(3)


def blob_with(t):
    # make_blob packs (params.to_c(meter_rect), template): the same with another template
    orig = _engine.load_template
    _engine.load_template = lambda p: np.ascontiguousarray(t)
    try:
        return _engine.make_blob(params)
    finally:
        _engine.load_template = orig


templates = {'real': tpl, 'flat 128 (A = 0)': np.full_like(tpl, 128), 'random': rng.integers(0, 256, tpl.shape, dtype=np.uint8),
             'real - mean + 128': np.clip(tpl.astype(int) - int(tpl.mean()) + 128, 0, 255).astype(np.uint8)}
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
(H, W) = base.shape[1:3]
B = 1024
g = torch.Generator(device=dev)
g.manual_seed(1)
frame_sets = {'synthetic (bench)': bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), B, 2024, dev),
              'all 128 (B = 0)': torch.full((B, H, W, 3), 128, dtype=torch.uint8, device=dev),
              'all 0': torch.zeros((B, H, W, 3), dtype=torch.uint8, device=dev),
              'random bytes': torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev, generator=g)}
stream = torch.cuda.current_stream().cuda_stream
print('template mean %.1f, min %d, max %d' % (tpl.mean(), tpl.min(), tpl.max()))
# the first launches of a process run ~5 % slower than the later ones: 300 untimed calls first, as bench.py does
heat = _hip.Context(blob_real, 0)
for _ in range(300):
    heat.process_batch_dev(frame_sets['synthetic (bench)'].data_ptr(), B, H, W, want_host=False, stream=stream)
torch.cuda.synchronize()
for (tn, t) in templates.items():
    ctx = _hip.Context(blob_with(t), 0)
    row = []
    for (fn, fr) in frame_sets.items():
        for _ in range(240):   # (a context's creation leaves the chip idle; its clocks take tens of milliseconds to come back)
            ctx.process_batch_dev(fr.data_ptr(), B, H, W, want_host=False, stream=stream)
        torch.cuda.synchronize()
        ctx.set_profiling(2)
        ctx.timings()
        for _ in range(40):
            ctx.process_batch_dev(fr.data_ptr(), B, H, W, want_host=False, stream=stream)
        torch.cuda.synchronize()
        (ms, cnt) = ctx.timings()['k_match']
        ctx.set_profiling(0)
        row.append('%s %.1f us' % (fn, ms / cnt * 1e3))
    print('template %-20s | %s   [%s]' % (tn, ' | '.join(row), ctx.last_match().get('layout')), flush=True)
    ctx.close() if hasattr(ctx, 'close') else None
