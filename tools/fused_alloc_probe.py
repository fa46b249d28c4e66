#!/usr/bin/env python3
"""Does the fused-mask launch's time depend on WHERE its buffers lie, or on when it runs?  One process, config 5 (B = 512,
1080p): several allocations of the same buffers (freed and re-made, a pad of another size in between), each timed three times
over 16 launches.  (Round 6: two runs of tools/run_stage.py a minute apart differed by 10 % on one box.)
With the diagnostic build (MELF_LIB_PATH=meterelf_amd/csrc/libmeterelf_hip_diag.so) the bare 3:1 stream of melf_stream_probe_dev
runs over the SAME buffers (static split and queue2): does the memory system itself see the placement?  Then ONE block holding
both buffers, the masks at several distances behind the frames.
    python3 tools/fused_alloc_probe.py [trials]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from meterelf_amd import _engine, _hip, _params

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = _hip.Context(_engine.make_blob(_params.load(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml'))), 0)
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
(B, H, W) = (512, 1080, 1920)
g = torch.Generator(device=dev)
g.manual_seed(7)
pads = []
for t in range(trials):
    frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
    masks = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    ctx.set_profiling(False)
    for _ in range(4):
        ctx.hls_inrange_close_dev(frames.data_ptr(), B, H, W, masks.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    line = []
    for rep in range(3):
        ctx.set_profiling(True)
        ctx.timings()
        for _ in range(16):
            ctx.hls_inrange_close_dev(frames.data_ptr(), B, H, W, masks.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        (ms, n) = ctx.timings()['k_fused_mask']
        line.append(ms / n)
    bare = ''
    if hasattr(_hip.lib(), 'melf_stream_probe_dev'):
        res = []
        for ch in (0, 2):
            ctx.set_profiling(True)
            ctx.timings()
            for _ in range(16):
                ctx.stream_probe_dev(frames.data_ptr(), B * H * W * 3, masks.data_ptr(), ch, stream=stream)
            torch.cuda.synchronize()
            (ms, n) = ctx.timings()['k_stream_probe']
            res.append(ms / n)
        bare = ' | bare stream static %.4f queue2 %.4f ms' % tuple(res)
    print('allocation %d: frames at 0x%x, masks at 0x%x | ms per launch, 3 x 16 launches: %s | %.3f of 8 TB/s%s' % (
        t, frames.data_ptr(), masks.data_ptr(), ' '.join('%.4f' % v for v in line), B * H * W * 4 / (min(line) * 1e-3) / 8e12, bare), flush=True)
    del frames, masks
    torch.cuda.empty_cache()
    pads.append(torch.empty(((t * 37 + 11) << 20,), dtype=torch.uint8, device=dev))   # moves the next allocation somewhere else

# one block, the masks at different distances behind the frames
del pads
torch.cuda.empty_cache()
FB = B * H * W * 3
block = torch.empty((FB + B * H * W + (64 << 20),), dtype=torch.uint8, device=dev)
block[:FB] = torch.randint(0, 256, (FB,), dtype=torch.uint8, device=dev, generator=g)
for gap in (0, 4096, 1 << 20, (2 << 20) + 4096, 16 << 20, (33 << 20) + 512):
    mp = block.data_ptr() + FB + gap
    mp += (-mp) % 256
    ctx.set_profiling(False)
    for _ in range(4):
        ctx.hls_inrange_close_dev(block.data_ptr(), B, H, W, mp, stream=stream)
    torch.cuda.synchronize()
    line = []
    for rep in range(2):
        ctx.set_profiling(True)
        ctx.timings()
        for _ in range(16):
            ctx.hls_inrange_close_dev(block.data_ptr(), B, H, W, mp, stream=stream)
        torch.cuda.synchronize()
        (ms, n) = ctx.timings()['k_fused_mask']
        line.append(ms / n)
    print('one block at 0x%x: masks %10d bytes behind the frames: %s ms' % (block.data_ptr(), mp - block.data_ptr() - FB, ' '.join('%.4f' % v for v in line)), flush=True)
