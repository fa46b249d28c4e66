#!/usr/bin/env python3
"""k_match launch time over batch sizes and wave layouts (GPU box):   python3 tools/match_sweep.py [sample dir] [n ...]
For each batch size: the tuned kernel in the planner's layout, in every forced layout (MELF_MATCH_LAYOUT=rb,np for the
candidates around the planner's choice), and the general kernel.  Validates mfma_plan's cost model and the
tuned / general dispatch threshold of pick_match_kind.  Times are the dispatch's own start / stop stamps."""
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

args = sys.argv[1:]
sd = args[0] if args and not args[0].isdigit() else 'sample-images1'
sizes = [int(a) for a in args if a.isdigit()] or [64, 128, 192, 256, 288, 320, 384, 448, 480, 512, 576, 640, 704, 768, 832, 896, 960,
                                                   1024, 1056, 1088, 1120, 1536, 2048]
pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
params = _params.load(pfile)
blob = _engine.make_blob(params)
os.environ['MELF_MATCH'] = 'fast'
fast = _hip.Context(blob, 0)
os.environ['MELF_MATCH'] = 'gen'
gen = _hip.Context(blob, 0)
del os.environ['MELF_MATCH']
auto = _hip.Context(blob, 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
NMAX = max(sizes)
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), NMAX, 2024, dev)
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
P = auto.params
(crows, ccols) = (min(P.rect_y1, H) - min(P.rect_y0, H), min(P.rect_x1, W) - min(P.rect_x0, W))
rh = crows - P.th + 1


def timed(ctx, n, reps=12):
    for _ in range(3):
        ctx.process_batch_dev(frames.data_ptr(), n, H, W, want_host=False, stream=stream)
    torch.cuda.synchronize()
    ctx.set_profiling(2)
    ctx.timings()
    for _ in range(reps):
        ctx.process_batch_dev(frames.data_ptr(), n, H, W, want_host=False, stream=stream)
    torch.cuda.synchronize()
    (ms, cnt) = ctx.timings()['k_match']
    ctx.set_profiling(0)
    return ms / max(cnt, 1) * 1e3, ctx.last_match()


print('%s: crop %dx%d, map rows %d; k_match microseconds per launch' % (sd, ccols, crows, rh))
for n in sizes:
    row = []
    os.environ.pop('MELF_MATCH_LAYOUT', None)
    (t_auto, info) = timed(auto, n)
    q = _hip.match_layout_query(P.th, P.tw, crows, ccols, n)
    row.append('default=%s/%s %.1f' % (info['kernel'], info['layout'], t_auto))
    if q['kernel'] == 'mfma':
        (rb0, np0) = (q['rows_per_wave'], q['pair_waves'] // 2)
        row.append('plan rb%d np%d' % (rb0, np0))
        cands = set()
        for rb in (2, 3, 4, 5):
            cands.add((rb, 0))
            if rb < 5 and ccols - P.tw + 1 > 32:
                for np_ in range(0, rh // (2 * rb + 1) + 2):   # the fewest pairs that fit one / two rounds
                    na = max(0, -(-(rh - (2 * rb + 1) * np_) // rb))
                    w = (na + 2 * np_) * ((n + 31) // 32)
                    if w <= 1024 or (1024 < w <= 2048 and (na + 2 * (np_ - 1)) * ((n + 31) // 32) > 2048):
                        cands.add((rb, np_))
                        break
        for (rb, np_) in sorted(cands):
            os.environ['MELF_MATCH_LAYOUT'] = '%d,%d' % (rb, np_)
            (t, inf) = timed(fast, n)
            row.append('rb%d,%d(%dw)=%.1f' % (rb, np_, inf['waves'], t))
        # 4-row tiles in 2 / 4 K slices (round 5): the fewest pairs that fit one round
        for ks in (2, 4):
            for np_ in range(0, rh // 9 + 2):
                na = max(0, -(-(rh - 9 * np_) // 4))
                if (na + 2 * np_) * ((n + 31) // 32) * ks <= 1024 or np_ == rh // 9 + 1:
                    if ccols - P.tw + 1 <= 32 and np_ > 0:
                        break
                    os.environ['MELF_MATCH_LAYOUT'] = '4,%d,%d' % (np_, ks)
                    (t, inf) = timed(fast, n)
                    row.append('rb4,%d/k%d(%dw)=%.1f' % (np_, ks, inf['waves'], t))
                    break
        os.environ.pop('MELF_MATCH_LAYOUT', None)
    (t_gen, inf) = timed(gen, n)
    row.append('gen(%dw)=%.1f' % (inf['waves'], t_gen))
    print('n=%-5d %s' % (n, '  '.join(row)), flush=True)
