#!/usr/bin/env python3
"""Runs one stage repeatedly on GPU 0 -- the command to put behind
`rocprofv3 ... -- python3 tools/run_stage.py ...` for per-kernel traces and PMC passes.

    python3 tools/run_stage.py fused [--iters 20] [--batch 256]
    python3 tools/run_stage.py full  [--iters 10] [--batch 1024]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('stage', choices=['fused', 'full'])
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--sample-dir', default='sample-images1')
    ap.add_argument('--hw', default='480x640', help='fused stage: frame rows x columns')
    ap.add_argument('--profiling', type=int, default=1, help='full stage: 0 no event records, 1 every kernel, 2 only k_match (the bench setting)')
    ap.add_argument('--device-records', action='store_true', help='full stage: records stay in HBM, no copy and no sync per call (the bench loop)')
    ap.add_argument('--resident', action='store_true', help='full stage: melf_ctx_set_frames_resident(1): consecutive calls alternate between the two lanes')
    ap.add_argument('--nbuf', type=int, default=4, help='distinct buffer sets the launches rotate over (beyond the Infinity Cache)')
    a = ap.parse_args()
    import torch
    from meterelf_amd import _engine, _hip, _params
    pfile = os.path.join(ROOT, 'tests', 'golden', a.sample_dir, 'params.yml')
    ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    ctx.set_profiling(True)
    if a.stage == 'fused':
        B = a.batch or 256
        (H, W) = (int(v) for v in a.hw.split('x'))
        NB = max(1, a.nbuf)
        frames = torch.randint(0, 256, (NB * B, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
        masks = torch.empty((NB * B, H, W), dtype=torch.uint8, device=dev)

        def launch(i):
            b = i % NB
            ctx.hls_inrange_close_dev(frames.data_ptr() + b * B * H * W * 3, B, H, W, masks.data_ptr() + b * B * H * W, stream=stream)
        ctx.set_profiling(False)
        for i in range(10):  # untimed warm-up launches
            launch(i)
        torch.cuda.synchronize()
        ctx.set_profiling(bool(a.profiling))
        for i in range(a.iters):
            launch(i)
        torch.cuda.synchronize()
    else:
        import bench
        import glob
        import numpy as np
        from meterelf_amd._image import imread_bgr
        B = a.batch or 1024
        files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', a.sample_dir, '*.jpg')))
                 if os.path.basename(f) not in bench.REJECTED]
        base = np.stack([imread_bgr(f) for f in files if imread_bgr(f).shape == imread_bgr(files[0]).shape])
        NB = max(1, a.nbuf)
        frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), NB * B, 2024, dev)
        (H, W) = base.shape[1:3]
        ctx.set_profiling(a.profiling)
        ctx.set_frames_resident(a.resident)
        recs = torch.empty((NB, B * _hip.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        for i in range(a.iters):
            if a.device_records:
                ctx.process_batch_dev(frames.data_ptr() + (i % NB) * B * H * W * 3, B, H, W, want_host=False, stream=stream,
                                      d_results_ptr=recs[i % NB].data_ptr())
            else:
                ctx.process_batch_dev(frames.data_ptr() + (i % NB) * B * H * W * 3, B, H, W, want_host=True, stream=stream)
        torch.cuda.synchronize()
    t = ctx.timings()
    for (k, (ms, n)) in t.items():
        if n:
            print('%-14s launches %4d  avg %.4f ms' % (k, n, ms / n))


if __name__ == '__main__':
    main()
