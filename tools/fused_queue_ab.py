#!/usr/bin/env python3
"""A/B of the fused-mask launch's work queue ON THE SAME BUFFERS (diagnostic build): round 6 found that the launch's time
depends on where its buffers lie (tools/fused_alloc_probe.py: +-5 % between allocations of one process, stable within one), so
the round-5 A/Bs, which compared separate processes, could not see a few per cent.  Per allocation: the static split, each
(MELF_FUSED_DYN, MELF_FUSED_BIG) variant, the static split again; 16 launches each.
    MELF_LIB_PATH=meterelf_amd/csrc/libmeterelf_hip_diag.so python3 tools/fused_queue_ab.py [allocations] [HxW] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_diag.so'))
import torch

from meterelf_amd import _engine, _hip, _params

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 5
(H, W) = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '1080x1920').split('x'))
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
NB = 4 if B * H * W * 4 < (1 << 30) else 1     # small workloads rotate over buffers beyond the Infinity Cache
VARIANTS = [(0, 0)] + [tuple(int(x) for x in v.split(',')) for v in os.environ.get('VARIANTS', '2,0 3,0 4,0 8,0 2,50 4,70 4,80 8,90').split()] + [(0, 0)]
ctx = _hip.Context(_engine.make_blob(_params.load(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml'))), 0)
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev)
g.manual_seed(7)
pads = []
ref = None
print('%dx%d B=%d, %d buffer set(s); columns: %s' % (H, W, B, NB, ' '.join('%d,%d' % v for v in VARIANTS)))
for t in range(trials):
    frames = torch.randint(0, 256, (NB * B, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
    masks = torch.empty((NB * B, H, W), dtype=torch.uint8, device=dev)

    def launch(i):
        b = i % NB
        ctx.hls_inrange_close_dev(frames.data_ptr() + b * B * H * W * 3, B, H, W, masks.data_ptr() + b * B * H * W, stream=stream)
    out = []
    for (dyn, big) in VARIANTS:
        os.environ['MELF_FUSED_DYN'] = str(dyn)
        os.environ['MELF_FUSED_BIG'] = str(big)
        ctx.set_profiling(False)
        for i in range(4):
            launch(i)
        torch.cuda.synchronize()
        ctx.set_profiling(True)
        ctx.timings()
        for i in range(16):
            launch(i)
        torch.cuda.synchronize()
        (ms, n) = ctx.timings()['k_fused_mask']
        out.append(ms / n)
        chk = int(masks[:B].to(torch.int64).sum().item())
        if ref is None or t != ref[0]:
            ref = (t, chk)
        assert chk == ref[1], 'masks differ'
    base = min(out[0], out[-1])
    print('allocation %d: %s | best variant %.3f of the static split' % (t, ' '.join('%.4f' % v for v in out), min(out[1:-1]) / base), flush=True)
    del frames, masks
    torch.cuda.empty_cache()
    pads.append(torch.empty(((t * 37 + 11) << 20,), dtype=torch.uint8, device=dev))
