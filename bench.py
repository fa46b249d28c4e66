#!/usr/bin/env python3
"""bench.py -- frames/sec of the meterelf hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the whole per-frame path (template match -> dial reading
-> digits, BASELINE config "Batch=1024, full pipeline, 4 dials") over one batch
of synthetic 640x480 frames that are already resident in HBM.  Each rank owns
one GPU and its own batch (weak scaling); the only collective is the RCCL
broadcast of the calibration blob at set-up.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline       dominant kernel of the step (k_match), hipEvent-timed inside
                 the library on the stream it runs on
  fused_mask     BASELINE config 2 (B=256, fused HLS+inRange+closing kernel) with
                 its own HBM roofline
  cpu_baseline   the CPU oracle (restated port, 1 thread) on a bounded sample of
                 the same frames; the same sample doubles as an in-run parity gate
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REJECTED = ('20180814021309-01-e01.jpg', '20180814021310-00-e02.jpg')
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
I8_MFMA_PEAK_TOPS = 5000.0     # dense i8 MFMA = 2x the ~2.5 PF bf16 rate (same guide, Matrix cores)


def synth_frames_gpu(torch, base_u8, n, seed, device, shift=8, sigma=2.0):
    """BASELINE config 3 synthesis: fixture (i mod K) circularly
    shifted by (dx, dy) in [-shift, shift]^2 plus N(0, sigma^2) integer noise."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    K = base_u8.shape[0]
    out = torch.empty((n,) + tuple(base_u8.shape[1:]), dtype=torch.uint8, device=device)
    shifts = rng.integers(-shift, shift + 1, size=(n, 2))
    for i in range(n):
        img = torch.roll(base_u8[i % K], shifts=(int(shifts[i, 1]), int(shifts[i, 0])), dims=(0, 1)).to(torch.float32)
        img += torch.round(torch.randn(img.shape, generator=gen, device=device) * sigma)
        out[i] = img.clamp_(0, 255).to(torch.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='frames per GPU per step')
    ap.add_argument('--sample-dir', default='sample-images1')
    ap.add_argument('--cpu-sample', type=int, default=256, help='frames timed through the CPU oracle (0 = skip)')
    ap.add_argument('--no-fused-mask', action='store_true')
    ap.add_argument('--overlap', action='store_true',
                    help='the K steps as ONE melf_process_stream_dev call (steps overlap on two lanes: more frames/s, but the '
                         'match kernel then shares the SIMDs and its own launch time -- the roofline figure -- stretches)')
    ap.add_argument('--no-jpeg', action='store_true', help='skip the JPEG-files-in block (SURVEY 8 f1)')
    args = ap.parse_args()

    import torch
    from meterelf_amd import _engine, _hip, _params
    from meterelf_amd._image import imread_bgr

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs a torch.distributed.run launch with %d ranks' % (args.gpus, args.gpus))
    if not torch.cuda.is_available() or _hip.device_count() < 1:
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    pfile = os.path.join(GOLDEN, args.sample_dir, 'params.yml')
    dist = None
    if world > 1 or 'RANK' in os.environ:   # launched by torch.distributed.run (also with one rank)
        from meterelf_amd import _dist
        dist = _dist.init_process_group('nccl')
        (blob, names) = (None, None)
        if rank == 0:
            params = _params.load(pfile)
            blob, names = _engine.make_blob(params), params.dial_names
        (blob, dev_blob, names) = _dist.broadcast_blob(blob, names, src=0, device=device)
        ctx = _hip.Context(blob, local_rank, blob_device_ptr=dev_blob.data_ptr())
    else:
        params = _params.load(pfile)
        blob = _engine.make_blob(params)
        ctx = _hip.Context(blob, local_rank)
    P = ctx.params

    # ---- synthetic workload, resident in HBM ----
    files = [f for f in sorted(glob.glob(os.path.join(GOLDEN, args.sample_dir, '*.jpg')))
             if os.path.basename(f) not in REJECTED]
    base = [imread_bgr(f) for f in files]
    shape = base[0].shape
    base = np.stack([b for b in base if b.shape == shape])
    (H, W) = shape[:2]
    base_gpu = torch.from_numpy(base).to(device)
    B = args.batch
    frames = synth_frames_gpu(torch, base_gpu, B, 2024 + rank, device)
    del base_gpu
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    # results stay on the device during the timed region (one record buffer per step would do the
    # same; the path has no step-to-step dependency) and are copied to the host once at the end
    d_results = torch.empty(B * _hip.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=device)

    def step():
        ctx.process_batch_dev(frames.data_ptr(), B, H, W, d_results_ptr=d_results.data_ptr(), want_host=False,
                              stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # A few steps with every kernel bracketed by events (informational per-kernel times), then the timed
    # region with events around the dominant kernel only: each event record is a barrier packet in the
    # queue, eight of them per step cost ~4 % of the step.
    ctx.set_profiling(1)
    ctx.timings()
    for _ in range(max(3, args.warmup)):
        step()
    torch.cuda.synchronize()
    kt_all = ctx.timings()
    ctx.set_profiling(2)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not args.overlap:
        for _ in range(args.steps):
            step()
    else:
        # the K steps as one stream of K batches (melf_process_stream_dev): consecutive steps alternate between the
        # context's two pipeline lanes, so one step's prep / dials kernels run in the tail of the other's match kernel
        ctx.process_stream_dev(frames.data_ptr(), args.steps, 0, B, H, W, d_results.data_ptr(), 0, stream=stream)
    recs = d_results.cpu().numpy().view(_hip.RESULT_DTYPE)   # D2H of the last step's records: inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kt = ctx.timings()
    ctx.set_profiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_ok = int((recs['status'] == 0).sum())

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel (k_match) ----
    crows = min(P.rect_y1, H) - min(P.rect_y0, H)
    ccols = min(P.rect_x1, W) - min(P.rect_x0, W)
    positions = (crows - P.th + 1) * (ccols - P.tw + 1)
    mac_per_frame = positions * P.th * P.tw            # 186 045 552 for sample-images1
    (match_ms, match_n) = kt['k_match']
    (dials_ms, dials_n) = kt_all['k_dials']
    (prep_ms, prep_n) = kt_all['k_lplane']
    match_avg_ms = match_ms / max(match_n, 1)
    frames_per_launch = B * args.steps / max(match_n, 1)   # a step may issue its match as several launches (pipeline lanes)
    tops = 2.0 * mac_per_frame * frames_per_launch / (match_avg_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as fp:
            traffic = json.load(fp)
    roofline = {
        'kernel': 'k_match', 'bound': 'mfma', 'achieved': round(tops, 3), 'peak': I8_MFMA_PEAK_TOPS,
        'unit': 'TFLOP/s', 'frac': round(tops / I8_MFMA_PEAK_TOPS, 5),
        'traffic': (traffic or {}).get('k_match'),
        'avg_launch_ms': round(match_avg_ms, 4), 'launches': match_n,
        'algorithmic': '%d int MAC/frame x %d frames/launch, 2 ops per MAC' % (mac_per_frame, frames_per_launch),
        'note': 'exact integer TM_CCOEFF on v_mfma_i32_32x32x32_i8; algorithmic MACs (the Toeplitz form issues 1.28x as many), priced against the dense i8 MFMA peak',
        'k_dials_avg_launch_ms': round(dials_ms / max(dials_n, 1), 4),   # these two from the untimed all-kernel pass
        'k_prep_avg_launch_ms': round(prep_ms / max(prep_n, 1), 4),
    }

    # ---- BASELINE config 2: fused HLS + inRange + closing, B=256 ----
    fused = None
    if not args.no_fused_mask:
        FB = 256
        g = torch.Generator(device=device)
        g.manual_seed(1234)
        fframes = torch.randint(0, 256, (FB, 640, 480, 3), dtype=torch.uint8, device=device, generator=g)
        fmasks = torch.empty((FB, 640, 480), dtype=torch.uint8, device=device)
        for _ in range(args.warmup):
            ctx.hls_inrange_close_dev(fframes.data_ptr(), FB, 640, 480, fmasks.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        ctx.set_profiling(True)
        ctx.timings()
        tf0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.hls_inrange_close_dev(fframes.data_ptr(), FB, 640, 480, fmasks.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf0
        (fms, fn) = ctx.timings()['k_fused_mask']
        ctx.set_profiling(False)
        favg = fms / max(fn, 1)
        alg_bytes = FB * 640 * 480 * 4   # 3 B/px read + 1 B/px written
        gbs = alg_bytes / (favg * 1e-3) / 1e9
        fused = {
            'workload': 'B=256 640x480 uniform-random u8 frames, fused HLS+inRange+closing only',
            'frames_per_s': round(FB * args.steps / tf, 1),
            'roofline': {'kernel': 'k_fused_mask', 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                         'traffic': (traffic or {}).get('k_fused_mask'),
                         'avg_launch_ms': round(favg, 4), 'launches': fn,
                         'algorithmic': '1 228 800 B/frame x 256 frames/launch'},
        }
        del fframes, fmasks

    # ---- CPU baseline (oracle port) + in-run parity gate on the same sample ----
    cpu = None
    if world == 1 and args.cpu_sample > 0:
        from oracle import pyoracle as po
        S = min(args.cpu_sample, B)
        sample = frames[:S].cpu().numpy()
        op = po.Params(pfile)
        po.process_frames(sample[:2], op)  # warm the library
        tc0 = time.perf_counter()
        ores = po.process_frames(sample, op)
        tc = time.perf_counter() - tc0
        mism = 0
        for i in range(S):
            (r, o) = (recs[i], ores[i])
            same = int(r['status']) == o.status and int(r['match_x']) == o.match_x and int(r['match_y']) == o.match_y
            if same and o.status == 0:
                same = '{:07.3f}'.format(float(r['value'])) == '{:07.3f}'.format(o.value)
            mism += 0 if same else 1
        # the same sample again on every host core (threads over chunks; the C call releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        ncores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
        ncores = min(ncores, 16, S)  # one GPU's share of the host
        parts = [sample[i::ncores] for i in range(ncores) if len(sample[i::ncores])]
        with ThreadPoolExecutor(max_workers=ncores) as pool:
            tm0 = time.perf_counter()
            list(pool.map(lambda part: po.process_frames(part, op), parts))
            tm = time.perf_counter() - tm0
        cpu = {'value': round(S / tc, 2), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
               'all_cores': {'value': round(S / tm, 2), 'unit': 'frames/s', 'cores': ncores},
               'sample': 'first %d frames of the same batch through oracle/melf_oracle.c (exact direct '
                         'correlation, single thread; the reference itself needs OpenCV 3.4.5, absent here)' % S,
               'parity_mismatches_vs_gpu': mism}

    # ---- the same path fed with JPEG files (decode on the GPU), fixture files tiled to B ----
    jpeg = None
    if world == 1 and not args.no_jpeg:
        import glob as _glob
        jfiles = [f for f in sorted(_glob.glob(os.path.join(ROOT, 'tests', 'golden', args.sample_dir, '*.jpg')))
                  if os.path.basename(f) not in REJECTED]
        blobs = [open(f, 'rb').read() for f in jfiles]
        blobs = [b for b in blobs if _hip.jpeg_probe(b)[:3] == (H, W, True)]
        if blobs:
            JB = 1024
            batch = [blobs[i % len(blobs)] for i in range(JB)]
            (jf, jst) = ctx.jpeg_decode(blobs[:8], H, W)
            from meterelf_amd._image import imread_bgr
            same = bool((jst == 0).all()) and all(
                np.array_equal(jf[i], imread_bgr(f)) for (i, f) in enumerate([f for f in jfiles][:8])
                if _hip.jpeg_probe(open(f, 'rb').read())[:2] == (H, W))
            ctx.jpeg_process_batch(batch, H, W)  # warm-up (allocations)
            ctx.set_profiling(True)
            ctx.timings()
            tj0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                (jrecs, jstatus) = ctx.jpeg_process_batch(batch, H, W)
            tj = (time.perf_counter() - tj0) / reps
            jt = ctx.timings()
            ctx.set_profiling(False)
            jpeg = {'workload': '%d JPEG files (%d distinct %s fixtures, %.1f KB average) -> decode + full reading path, '
                                'file bytes in host memory to result records' % (JB, len(blobs), args.sample_dir,
                                                                                 sum(map(len, blobs)) / len(blobs) / 1024),
                    'files_per_s': round(JB / tj, 1), 'ms_per_call': round(tj * 1e3, 3),
                    'kernel_ms': {k: round(ms / c, 4) for (k, (ms, c)) in jt.items() if c and k.startswith('k_jpeg')},
                    'decoded_frames_equal_libjpeg_turbo': same, 'files_ok': int((jstatus == 0).sum())}

    line = {
        'metric': 'frames/sec (640x480), full pipeline, digits identical to the oracle',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': 'Batch=%d per GPU, full pipeline incl. TM_CCOEFF match + needle reading, 4 dials/frame, '
                               '%s params, frames %dx%d synthesised from the readable fixtures (shift +-8, noise sigma 2)'
                               % (B, args.sample_dir, W, H),
                   'global_batch': B * world, 'parallelism': 'dp%d' % world, 'frames_read_ok_last_step': n_ok},
        'roofline': roofline, 'cpu_baseline': cpu, 'fused_mask': fused, 'jpeg_decode': jpeg,
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
