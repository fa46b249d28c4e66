#!/usr/bin/env python3
"""k_match_gen tile-shape sweep on the full path (GPU box), one process, warm GPU: a fresh context per shape
(MELF_GEN_SHAPE=rows,colblocks,slices is read when a plan is made), k_match launch time from the dispatch's own stamps.
    python3 tools/gen_shape_sweep.py [sample dir] [batch]"""
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

sd = sys.argv[1] if len(sys.argv) > 1 else 'sample-images2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
blob = _engine.make_blob(_params.load(pfile))
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
NB = 4
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), NB * n, 2025, dev)
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
# reference records: the VALU kernel (an independent formulation) on batch 0
os.environ['MELF_MATCH'] = 'dot4'
ctx = _hip.Context(blob, 0)
ref = ctx.process_batch_dev(frames.data_ptr(), n, H, W, want_host=True, stream=stream).tobytes()
ctx.close()
os.environ['MELF_MATCH'] = 'gen'
nxs = (1, 2) if len(sys.argv) > 3 and sys.argv[3] == 'nx2' else (1,)
# slices = waves of the tile's workgroup: up to 8 for the small tile shapes (two waves per SIMD), 4 otherwise; 8 x 2 is not built
shapes = ['default'] + ['%d,%d,%d' % (rc, nx, ns) for rc in (2, 4, 6, 8) for nx in nxs for ns in range(1, 9)
                        if ns <= (8 if rc * nx <= 4 else 4) and (rc, nx) != (8, 2)]
for rep in range(int(os.environ.get('SWEEP_PASSES', '1'))):
    for sh in shapes:
        if sh == 'default':
            os.environ.pop('MELF_GEN_SHAPE', None)
        else:
            os.environ['MELF_GEN_SHAPE'] = sh
        ctx = _hip.Context(blob, 0)
        for i in range(12):
            ctx.process_batch_dev(frames.data_ptr() + (i % NB) * n * H * W * 3, n, H, W, want_host=False, stream=stream)
        torch.cuda.synchronize()
        recs = ctx.process_batch_dev(frames.data_ptr(), n, H, W, want_host=True, stream=stream)
        ok = recs.tobytes() == ref
        ctx.set_profiling(2)
        ctx.timings()
        for i in range(60):
            ctx.process_batch_dev(frames.data_ptr() + (i % NB) * n * H * W * 3, n, H, W, want_host=False, stream=stream)
        torch.cuda.synchronize()
        (ms, cnt) = ctx.timings()['k_match']
        info = ctx.last_match()
        print('pass %d %-10s k_match %.4f ms  %-12s waves %5d tiles %d  records %s' % (rep, sh, ms / cnt, info['layout'], info['waves'], info['tiles'],
                                                                                  'same as the VALU kernel\'s' if ok else 'DIFFER'), flush=True)
        ctx.close()
