#!/usr/bin/env python3
"""The fused-mask launch right after a process (or an idle stretch) starts: per-launch event times of the first N launches -- the
chip needs 20-30 ms of load before its clocks have settled (launches 4-8 are the slowest), which is why bench.py preheats.
    python3 tools/fused_first_launches.py [queue|static] [HxW] [batch] [launches]      (static: diagnostic build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from meterelf_amd import _engine, _hip, _params
mode = sys.argv[1] if len(sys.argv) > 1 else 'queue'
if mode == 'static':
    os.environ['MELF_FUSED_DYN'] = '0'
ctx = _hip.Context(_engine.make_blob(_params.load(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml'))), 0)
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
(H, W) = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '1080x1920').split('x'))
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
N = int(sys.argv[4]) if len(sys.argv) > 4 else 120
g = torch.Generator(device=dev); g.manual_seed(7)
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
masks = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
out = []
for i in range(N):
    ctx.set_profiling(True); ctx.timings()
    ctx.hls_inrange_close_dev(frames.data_ptr(), B, H, W, masks.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    (ms, n) = ctx.timings()['k_fused_mask']
    out.append(ms / n)
step = max(10, N // 12)
print(mode, '%dx%d B=%d' % (H, W, B), 'first launches ms:', ' '.join('%.4f' % v for v in out[:12]), '... | means of %d:' % step, ' '.join('%.4f' % (sum(out[k:k + step]) / step) for k in range(0, N - step + 1, step)))
