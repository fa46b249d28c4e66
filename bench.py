#!/usr/bin/env python3
"""bench.py -- frames/sec of the meterelf hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torchrun environment the script launches its own N ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process; the parent never touches
the GPU) -- or run it under torch.distributed.run yourself.  One rank per GPU, rank 0 prints ONE JSON line.

A step = one pass of the whole per-frame path (template match -> dial reading -> digits, BASELINE config 3
"Batch=1024, full pipeline, 4 dials") over one batch of synthetic 640x480 frames already resident in HBM.
Consecutive steps rotate over --nbuf (default 4) DISTINCT batches (3.8 GB of frames), so the input of a step is
never what the previous step left in the 256 MB Infinity Cache.  Each rank owns one GPU and its own batches (weak
scaling); the only collectives are set-up ones (RCCL broadcast of the calibration blob) and the timing barrier.

Objects on the line besides the contract's fields:
  roofline       dominant kernel of the step (k_match), timed by the dispatch's own start/stop stamps
                 (hipExtLaunchKernelGGL) on the stream it runs on, over the timed region
  single_lane    the same K steps on ONE caller stream without the hint (every kernel behind the previous one): the launches roofline /
                 kernel_ms describe; the timed region itself runs on the context's two lanes (--mode resident | two_streams)
  sustained      >= 2 s of back-to-back steps (DVFS-settled rate), and the same length on one lane with k_match's stamps
  two_streams    ~1 s of the steps alternating between two caller streams
  resident_hint  ~1 s on ONE caller stream with melf_ctx_set_frames_resident (the library alternates its lanes per call)
  cpu_baseline   the CPU oracle (restated port) on a bounded sample of the same frames; doubles as parity gate
  fused_mask     BASELINE config 2 (B=256, fused HLS+inRange+closing) rotating over 4 buffer pairs, HBM roofline
  config4        BASELINE config 4 per GPU: sample-images2 params, 1024 frames/GPU, blob via RCCL broadcast
  config5        BASELINE config 5 per GPU: 1080p, 6 dials, 512 frames/GPU: fused mask (HBM roofline) + full path
  host_fed       the host-pointer entry point (frames in host memory, PCIe inclusive) -- never `value`
  jpeg_decode    the same path fed with JPEG files (decode on the GPU)
  rccl_ranks     all_reduce(1) over the nccl (= RCCL) backend: number of ranks that took part
Every timed block runs untimed launches first (`untimed_preheat_steps` / `untimed_preheat_launches` on the line: 300 steps in
front of the headline, 60 ms of launches in front of the fused-mask and config-5 blocks): the chip's clocks settle 20-30 ms after
a load begins (tools/fused_first_launches.py), and W + K launches right behind a buffer allocation sit inside that.
"""
import argparse
import glob
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

# Streams only overlap on the GPU if they sit on different hardware queues, and the HIP runtime spreads ALL of a process's
# streams (torch creates pools of them) over GPU_MAX_HW_QUEUES = 4 queues by default: with four, the two caller streams of
# the `two_streams` block shared a queue on some runs and did not overlap at all.  Must be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REJECTED = ('20180814021309-01-e01.jpg', '20180814021310-00-e02.jpg')
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
I8_MFMA_PEAK_TOPS = 5000.0     # dense i8 MFMA = 2x the ~2.5 PF bf16 rate (same guide, Matrix cores)
ALL_BLOCKS = ('sustained', 'twostream', 'fused', 'config4', 'config5', 'cpu', 'hostfed', 'jpeg')


# ------------------------------------------------------------------ launcher ----
def launch_ranks(args):
    """--gpus N without a torchrun environment: start the N ranks as a child process tree.  Nothing in this
    (parent) process has touched HIP or torch.cuda; it only waits and passes the exit code on."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------- workloads ----
def synth_frames_gpu(torch, base_u8, n, seed, device, shift=8, sigma=2.0):
    """BASELINE config 3/4 synthesis: fixture (i mod K) circularly shifted by (dx, dy) in
    [-shift, shift]^2 plus N(0, sigma^2) integer noise, clipped to u8."""
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


def load_fixture_frames(sample_dir):
    from meterelf_amd._image import imread_bgr
    files = [f for f in sorted(glob.glob(os.path.join(GOLDEN, sample_dir, '*.jpg')))
             if os.path.basename(f) not in REJECTED]
    base = [imread_bgr(f) for f in files]
    shape = base[0].shape
    return np.stack([b for b in base if b.shape == shape])


def config5_params_dir():
    """BASELINE config 5: 1920x1080 frames, meter_rect 250x250 inside the frame, six needle_data
    entries (the four of sample-images1 + two more at slightly moved centres), same template."""
    import yaml
    src = os.path.join(GOLDEN, 'sample-images1')
    with open(os.path.join(src, 'params.yml')) as fp:
        data = yaml.safe_load(fp)
    data['meter_rect'] = {'top_left': [1210, 420], 'bottom_right': [1460, 670]}
    extra = []
    for (k, nd) in enumerate(data['needle_data'][:2]):
        nd2 = dict(nd)
        nd2['name'] = '1.%d' % k
        nd2['center'] = [nd['center'][0] + 0.4, nd['center'][1] - 0.3]
        extra.append(nd2)
    data['needle_data'] = data['needle_data'] + extra
    d = tempfile.mkdtemp(prefix='melf_cfg5_')
    with open(os.path.join(d, 'params.yml'), 'w') as fp:
        yaml.safe_dump(data, fp)
    shutil.copy(os.path.join(src, 'dials_gray.png'), os.path.join(d, 'dials_gray.png'))
    return d


class Env:
    """Per-rank state: device, stream, process group."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.under_torchrun = 'RANK' in os.environ
        self.dist = None
        self.backend = None
        from meterelf_amd import _hip
        ndev = _hip.device_count()
        if not torch.cuda.is_available() or ndev < 1:
            raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
        if args.share_gpu:
            self.dev_index = self.local_rank % ndev   # rehearsal on a box with fewer GPUs than ranks (gloo only)
        else:
            if self.local_rank >= ndev:
                raise SystemExit('rank %d has no GPU: %d visible (use --share-gpu --backend gloo to rehearse)' % (self.rank, ndev))
            self.dev_index = self.local_rank
        torch.cuda.set_device(self.dev_index)
        self.device = torch.device('cuda', self.dev_index)
        if self.under_torchrun:
            import torch.distributed as dist
            self.backend = args.backend
            if args.share_gpu and self.backend == 'nccl' and self.world > ndev:
                raise SystemExit('--share-gpu puts several ranks on one GPU: RCCL cannot do that, use --backend gloo')
            dist.init_process_group(backend=self.backend)
            self.dist = dist
        # not torch's default stream (no implicit null-stream syncs).  With --streams 2 consecutive steps alternate between
        # two streams: the context runs them on its two pipeline lanes, so one step's prep / dials kernels overlap the
        # other's match kernel (steps are independent: different batches, different record slices)
        self.stream_objs = [torch.cuda.Stream(device=self.device) for _ in range(max(2, args.streams))]
        self.stream_obj = self.stream_objs[0]
        self.stream = self.stream_obj.cuda_stream

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def pinned_like(self, t):
        """A pinned host buffer of t's size (kept per size: the records of a timed region land here)."""
        key = int(t.numel())
        if not hasattr(self, '_pinned'):
            self._pinned = {}
        if key not in self._pinned:
            self._pinned[key] = self.torch.empty(key, dtype=self.torch.uint8, pin_memory=True)
        return self._pinned[key]

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def make_context(self, pfile):
        """Rank 0 reads params.yml and packs the calibration blob; with several ranks it is broadcast (RCCL with the
        nccl backend, GPU to GPU) and every rank creates its context from the received bytes."""
        from meterelf_amd import _engine, _hip, _params
        if self.dist is None:
            params = _params.load(pfile)
            return _hip.Context(_engine.make_blob(params), self.dev_index), list(params.dial_names)
        from meterelf_amd import _dist
        (blob, names) = (None, None)
        if self.rank == 0:
            params = _params.load(pfile)
            (blob, names) = (_engine.make_blob(params), list(params.dial_names))
        (blob, dev_blob, names) = _dist.broadcast_blob(blob, names, src=0, device=self.device if self.backend == 'nccl' else None)
        if dev_blob is not None:
            return _hip.Context(blob, self.dev_index, blob_device_ptr=dev_blob.data_ptr()), names
        return _hip.Context(blob, self.dev_index), names


def timed_steps(env, ctx, frames, B, nbuf, H, W, d_results, steps, frame_stride=None, nstreams=1):
    """Exactly `steps` steps between barrier + synchronize on both sides; step i reads batch i % nbuf and writes record slice
    i % (slices d_results holds).  nstreams = 2: consecutive steps alternate between two caller streams -- the context hands each
    stream one of its two lanes (own work buffers), so that a step's prep / dials kernels fill the other step's launch gaps and the
    tail of its match kernel; nothing orders the two streams against each other until the end of the region.  Returns (elapsed
    seconds of this rank, records of all slices)."""
    from meterelf_amd import _hip
    torch = env.torch
    fs = frame_stride or H * W * 3
    rsz = _hip.RESULT_DTYPE.itemsize
    nres = max(1, d_results.numel() // (B * rsz))
    env.barrier()
    env.sync()
    t0 = time.perf_counter()
    streams = env.stream_objs[:max(1, min(nstreams, len(env.stream_objs), nres))]
    for i in range(steps):
        b = i % nbuf
        ctx.process_batch_dev(frames.data_ptr() + b * B * fs, B, H, W, frame_stride=fs,
                              d_results_ptr=d_results.data_ptr() + (i % nres) * B * rsz, want_host=False,
                              stream=streams[i % len(streams)].cuda_stream)
    for so in streams[1:]:
        env.stream_obj.wait_stream(so)
    host = env.pinned_like(d_results)   # (allocated once per size, outside every timed region)
    with torch.cuda.stream(env.stream_obj):
        host.copy_(d_results, non_blocking=True)   # D2H of the records into pinned memory: inside the timed region
    env.sync()
    env.barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, host.numpy().copy().view(_hip.RESULT_DTYPE)


def step_event_times(env, ctx, frames, B, nbuf, H, W, d_results, steps, frame_stride=None):
    """The same `steps` steps once more with a hipEvent recorded on the stream after every step ("hipEvent
    timing, median reported").  Kept OUT of the contract's timed region: an event record is a barrier packet in the queue
    (a few microseconds between two steps).  Returns event-to-event milliseconds per step."""
    from meterelf_amd import _hip
    torch = env.torch
    fs = frame_stride or H * W * 3
    rsz = _hip.RESULT_DTYPE.itemsize
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    env.sync()
    with torch.cuda.stream(env.stream_obj):
        evs[0].record(env.stream_obj)
        for i in range(steps):
            b = i % nbuf
            ctx.process_batch_dev(frames.data_ptr() + b * B * fs, B, H, W, frame_stride=fs,
                                  d_results_ptr=d_results.data_ptr() + b * B * rsz, want_host=False, stream=env.stream)
            evs[i + 1].record(env.stream_obj)
    env.sync()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def percentiles(ms):
    a = np.sort(np.asarray(ms, dtype=np.float64))
    return {'n': int(len(a)), 'median_ms': round(float(np.median(a)), 4), 'p10_ms': round(float(np.percentile(a, 10)), 4),
            'p90_ms': round(float(np.percentile(a, 90)), 4), 'min_ms': round(float(a[0]), 4), 'max_ms': round(float(a[-1]), 4),
            'mean_ms': round(float(a.mean()), 4)}


def max_over_ranks(env, elapsed):
    """(max over ranks, list of every rank's seconds)"""
    if env.dist is None:
        return elapsed, [elapsed]
    torch = env.torch
    dev = env.device if env.backend == 'nccl' else torch.device('cpu')
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    allt = [torch.zeros_like(t) for _ in range(env.world)]
    env.dist.all_gather(allt, t)
    per = [float(x.item()) for x in allt]
    return max(per), per


def match_roofline(P, H, W, kt, frames_total, traffic_entry):
    crows = min(P.rect_y1, H) - min(P.rect_y0, H)
    ccols = min(P.rect_x1, W) - min(P.rect_x0, W)
    positions = (crows - P.th + 1) * (ccols - P.tw + 1)
    mac_per_frame = positions * P.th * P.tw            # 186 045 552 (sample-images1), 12 550 692 (sample-images2)
    (match_ms, match_n) = kt['k_match']
    match_avg_ms = match_ms / max(match_n, 1)
    frames_per_launch = frames_total / max(match_n, 1)
    tops = 2.0 * mac_per_frame * frames_per_launch / (match_avg_ms * 1e-3) / 1e12 if match_n else 0.0
    (traffic, source) = traffic_entry
    return {
        'kernel': 'k_match', 'bound': 'mfma', 'achieved': round(tops, 3), 'peak': I8_MFMA_PEAK_TOPS,
        'unit': 'TFLOP/s', 'frac': round(tops / I8_MFMA_PEAK_TOPS, 5), 'traffic': traffic, 'traffic_source': source,
        'avg_launch_ms': round(match_avg_ms, 4), 'launches': match_n,
        'algorithmic': '%d int MAC/frame x %d frames/launch, 2 ops per MAC' % (mac_per_frame, frames_per_launch),
        'note': 'exact integer TM_CCOEFF on v_mfma_i32_32x32x32_i8 (+ one 2:4-sparse v_smfmac_i32_32x32x64_i8 per template row and '
                'column block in the tuned kernel: the first and last Toeplitz blocks share it); algorithmic MACs only (Toeplitz zero '
                'padding is not counted), priced against the DENSE i8 MFMA peak.  The match kernels also add up the window sums of TM_CCOEFF '
                'themselves (k_match_mfma since round 3, k_match_gen since round 4; before, a separate column-sum launch of 8-15 us): '
                'their launches are 5-10 us longer for that and the step 8-14 us shorter.  The launch is power-limited and its clock follows the DATA: the same '
                'launch takes 102 us on frames of one value and 156 on real ones, which are as bad as random bytes (profiles/r06/match_operand_power.txt)',
    }


def dials_roofline(kernel_ms, traffic, label):
    """k_dials is bound by the vector units, not by bytes or matrix FLOPs -- instruction issue while a SIMD's four waves run
    together, the dependency chains of its last wave afterwards: its 'roofline' is the share of the vector units'
    cycles the launch keeps busy.  The counters come from the PMC pass of tools/profile_round.sh (rocprofv3 cannot run inside
    this process; traffic.json is stamped with the kernel sources it was measured on), the launch time from this run."""
    v = traffic.valu_entry(label + ':k_dials')
    out = {'kernel': 'k_dials', 'bound': 'valu', 'avg_launch_ms': kernel_ms.get('k_dials'), 'source': traffic.source,
           'note': 'avg_launch_ms: hipEvents around each launch in the per-kernel pass of this run (a few microseconds more than an '
                   'unbracketed launch); counters per launch from the PMC pass'}
    if v:
        out.update({'achieved': round(v['valu_busy_frac'], 4), 'peak': 1.0, 'unit': 'fraction of vector-unit cycles busy', 'frac': round(v['valu_busy_frac'], 4),
                    'insts_valu_per_launch': v.get('insts_valu_per_launch'), 'waves': v.get('waves'),
                    'formula': '4 x SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)'})
    else:
        out.update({'achieved': None, 'frac': None})
    return out


def jpeg_roofline(jt, traffic):
    """k_jpeg_huff, the kernel a JPEG call waits for, is bound by neither bytes nor arithmetic: a workgroup (one image) walks
    9-17 synchronisation rounds of 70-100 dependent decode steps, most of them on one or two waves (k_jpeg.hip, J1).  What
    can be said about it in roofline terms is how little of the vector units it uses; the chain itself is in the note."""
    (ms, c) = jt.get('k_jpeg_huff', (0.0, 0))
    out = {'kernel': 'k_jpeg_huff', 'bound': 'latency', 'launches_per_call': c, 'avg_launch_ms': round(ms / c, 4) if c else None,
           'note': 'a dependency chain, not a throughput kernel: the state-only decode step is 43 instructions (one LDS lookup, the next '
                   'stream dword requested a step ahead) and costs a lone wave ~415 cycles, ~650 with all eight waves of the workgroup; '
                   'an image needs 8-17 rounds of 70-100 such steps after the speculative pass (profiles/r06/jpeg_huff_rounds.txt); '
                   'six phase hypotheses in round 0 would not shorten the chain (model: profiles/r06/huff_hypotheses_model.txt)'}
    v = traffic.valu_entry('jpeg:k_jpeg_huff') if traffic else None
    if v:
        out.update({'achieved': round(v['valu_busy_frac'], 4), 'peak': 1.0, 'unit': 'fraction of vector-unit cycles busy', 'frac': round(v['valu_busy_frac'], 4),
                    'insts_valu_per_launch': v.get('insts_valu_per_launch'), 'waves': v.get('waves'), 'source': traffic.source,
                    'traffic': traffic.get('jpeg:k_jpeg_huff')[0],
                    'formula': '4 x SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)'})
    else:
        out.update({'achieved': None, 'frac': None, 'source': traffic.source if traffic else None})
    return out


def prep_roofline(P, H, W, kernel_ms, traffic, label, frames_per_launch):
    """k_prep_lplane is HBM-bound; priced on ALGORITHMIC bytes (SURVEY.md 8d's rule): per frame the meter_rect crop read once
    (3 B/px) + the L' plane written once in the match kernels' fragment order (1 B/px, rows padded to whole 32-column blocks) +
    the row-window sums written once (2 B per map column, padded to 32-column blocks).  The counter traffic of the PMC pass
    stands beside it as `traffic`: the ratio is what the kernel moves beyond the bytes it has to."""
    crows = min(P.rect_y1, H) - min(P.rect_y0, H)
    ccols = min(P.rect_x1, W) - min(P.rect_x0, W)
    nkb = (ccols + 31) // 32
    rwp = 32 * ((ccols - P.tw + 1 + 31) // 32)
    per_frame = crows * ccols * 3 + crows * nkb * 32 + crows * rwp * 2
    ms = kernel_ms.get('k_lplane')
    (tr, src) = traffic.get(label + ':k_prep_lplane')
    out = {'kernel': 'k_prep_lplane', 'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'avg_launch_ms': ms,
           'algorithmic': '%d B/frame (crop %d read + L plane %d + row-window sums %d written) x %d frames/launch' % (
               per_frame, crows * ccols * 3, crows * nkb * 32, crows * rwp * 2, frames_per_launch),
           'traffic': tr, 'traffic_source': src,
           'note': 'avg_launch_ms: hipEvents around each launch in the per-kernel pass of this run'}
    if ms:
        gbs = per_frame * frames_per_launch / (ms * 1e-3) / 1e9
        out.update({'achieved': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4)})
        if tr:
            out['traffic_over_algorithmic'] = round(tr / (per_frame * frames_per_launch), 3)
    else:
        out.update({'achieved': None, 'frac': None})
    return out


def kernel_sources_sha():
    """Hash of the kernel sources (k_*.hip and the device header): what the measured HBM traffic depends on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'meterelf_amd', 'csrc', 'k_*.hip')) +
                    [os.path.join(ROOT, 'meterelf_amd', 'csrc', 'melf_device.h')]):
        with open(f, 'rb') as fp:
            h.update(fp.read())
    return h.hexdigest()[:16]


class Traffic:
    """HBM bytes per launch from the PMC passes of tools/profile_round.sh (rocprofv3 cannot run inside this
    process): profiles/rNN/traffic.json = {"kernel_sources_sha16": ..., "per_launch_bytes": {"config3:k_match": N, ...},
    "files": [...]}.  The file records the kernel sources it was measured on; if they changed since, the figures are
    reported as stale (null) instead of being silently carried over."""

    def __init__(self):
        self.table = {}
        self.valu = {}
        self.source = None
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'traffic.json')), reverse=True)
        if not cands:
            return
        with open(cands[0]) as fp:
            t = json.load(fp)
        self.source = os.path.relpath(cands[0], ROOT)
        if t.get('kernel_sources_sha16') != kernel_sources_sha():
            self.source += ' (stale: kernel sources changed since those PMC passes)'
        else:
            self.table = t.get('per_launch_bytes', {})
            self.valu = t.get('valu', {})

    def get(self, key, default=None):
        return (self.table.get(key), self.source)

    def valu_entry(self, key):
        return self.valu.get(key)


def full_path_block(env, pfile, sample_dir, seed, steps, warmup, B, nbuf, sustained_s, traffic, cpu_sample, label, two_stream_s=0.0):
    """One context + nbuf distinct batches.  Three passes over the same steps:
      1. every kernel bracketed by events, one lane (informational per-kernel times: kernel_ms);
      2. `steps` steps on ONE lane with the dispatch's own stamps on k_match: the undisturbed launches `roofline` describes, and
         the step as the plain sum of its kernels (single_lane);
      3. the contract's timed region the way a throughput caller with frames in HBM drives the library: the same calls with
         consecutive calls alternating between TWO caller streams (--mode two_streams, the default: the context hands each stream one
         of its two lanes; nothing between the streams until the end of the region) or with melf_ctx_set_frames_resident on one caller
         stream (--mode resident: the library alternates its lanes itself, three event hand-overs per call -- 1-2 % faster at config 3,
         10-20 % slower at configs 4 and 5, whose kernels are short); no event records, no stamps.  Same records, byte for byte (checked).
    Then optional sustained / two-caller-stream runs and the CPU-oracle sample.  Returns a dict of raw results."""
    from meterelf_amd import _hip
    torch = env.torch
    (ctx, names) = env.make_context(pfile)
    two_lanes = not env.args.no_resident_hint
    ctx.set_frames_resident(False)
    P = ctx.params
    base = load_fixture_frames(sample_dir)
    (H, W) = base.shape[1:3]
    base_gpu = torch.from_numpy(base).to(env.device)
    frames = synth_frames_gpu(torch, base_gpu, B * nbuf, seed + env.rank, env.device)
    del base_gpu
    d_results = torch.zeros(B * nbuf * _hip.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=env.device)
    env.sync()

    def run(k, nstreams=1):
        return timed_steps(env, ctx, frames, B, nbuf, H, W, d_results, k, nstreams=nstreams)

    run(max(warmup, 1))
    # 1. a few steps with every kernel bracketed by events (informational per-kernel times) ...
    ctx.set_profiling(1)
    ctx.timings()
    run(max(3, nbuf))
    kt_all = ctx.timings()
    # 2. ... the same K steps on one lane with stamps on the dominant kernel only (the dispatch's own start / stop events: ~10 us
    # of idle time around that kernel per step), pre-heated: the event read-back above leaves the GPU idle for
    # a moment and the first launches after a pause run at a lower clock
    ctx.set_profiling(2)
    ns = env.args.streams
    preheat = env.args.preheat
    run(max(preheat, warmup, 2))
    ctx.timings()
    (elapsed1, recs1) = run(steps)
    kt = ctx.timings()
    (elapsed1_max, _p1) = max_over_ranks(env, elapsed1)
    # 3. the timed region
    ctx.set_profiling(0)
    resident = two_lanes and ns == 1 and env.args.mode == 'resident'
    ns_timed = 2 if (two_lanes and ns == 1 and not resident) else ns
    ctx.set_frames_resident(resident)
    run(max(preheat, warmup, 2), ns_timed)
    (elapsed, recs) = run(steps, ns_timed)
    (elapsed_max, per_rank) = max_over_ranks(env, elapsed)
    # per-step distribution: the same K steps again, an event after each (this rank's; single stream)
    step_ms = step_event_times(env, ctx, frames, B, nbuf, H, W, d_results, max(steps, 20))
    out = {'ctx': ctx, 'P': P, 'H': H, 'W': W, 'frames': frames, 'recs': recs, 'elapsed': elapsed_max,
           'per_rank_ms': [round(t / steps * 1e3, 4) for t in per_rank], 'kt': kt, 'kt_all': kt_all,
           'roofline': match_roofline(P, H, W, kt, B * steps, traffic.get(label + ':k_match'))}
    out['mode'] = ('one melf_process_batch_dev call per step on one caller stream with melf_ctx_set_frames_resident (the library alternates its two lanes)' if resident else
                   'one melf_process_batch_dev call per step, consecutive steps on %d caller stream(s)%s'
                   % (ns_timed, ' (the context runs them on its two lanes)' if ns_timed > 1 else ''))
    out['single_lane'] = {'ms_per_step': round(elapsed1_max / steps * 1e3, 4), 'frames_per_s': round(env.world * B * steps / elapsed1_max, 1), 'steps': steps,
                          'what': 'the same K steps as K melf_process_batch_dev calls on one stream: every kernel behind the previous one (the '
                                  'step is the sum of its kernels); these are the launches `roofline` and `kernel_ms` describe',
                          'records_identical_to_timed_region': bool(recs1.tobytes() == recs.tobytes())}
    out['step_events'] = dict(percentiles(step_ms), what='event-to-event time of each of %d further steps of the timed loop (one '
                              'hipEvent per step on the caller\'s stream, rank 0; the timed region itself carries no events)' % len(step_ms))
    out['match_layout'] = ctx.last_match()
    out['kernel_ms'] = {k: round(ms / n, 4) for (k, (ms, n)) in kt_all.items() if n}
    if sustained_s > 0:
        est = max(elapsed_max / steps, 1e-5)   # (the max over the ranks: every rank runs the same number of steps)
        k = int(sustained_s / est * 1.15) + nbuf
        (el, _r) = run(k, ns_timed)
        ctx.set_frames_resident(False)
        (el_max, _p) = max_over_ranks(env, el)
        out['sustained'] = {'seconds': round(el_max, 3), 'steps': k, 'ms_per_step': round(el_max / k * 1e3, 4),
                            'frames_per_s': round(env.world * B * k / el_max, 1)}
        # ... and the dominant kernel's settled launch time: the same length on one lane, stamps on k_match
        ctx.set_profiling(2)
        ctx.timings()
        (el1, _r) = run(k)
        kts = ctx.timings()
        ctx.set_profiling(0)
        (el1_max, _p) = max_over_ranks(env, el1)
        r = match_roofline(P, H, W, kts, B * k, (None, None))
        out['sustained'].update({'single_lane_ms_per_step': round(el1_max / k * 1e3, 4), 'k_match_avg_launch_ms': r['avg_launch_ms'], 'k_match_frac': r['frac']})
    ctx.set_frames_resident(False)
    if two_stream_s > 0 and ns == 1:
        # ~1 s of the steps alternating between two caller streams ...
        est = max(elapsed_max / steps, 1e-5)   # (the max over the ranks: every rank runs the same number of steps)
        k = int(two_stream_s / est * 1.2) + nbuf
        run(max(4, nbuf), 2)
        (el2, recs2) = run(k, 2)
        (el2_max, _p) = max_over_ranks(env, el2)
        # ... and on ONE caller stream with the frames-resident promise (melf_ctx_set_frames_resident: the library alternates its lanes
        # per call; three event hand-overs per call)
        ctx.set_frames_resident(True)
        run(max(4, nbuf))
        (el3, recs3) = run(k)
        ctx.set_frames_resident(False)
        (el3_max, _p) = max_over_ranks(env, el3)
        out['resident_hint'] = {'steps': k, 'seconds': round(el3_max, 3), 'ms_per_step': round(el3_max / k * 1e3, 4),
                                'frames_per_s': round(env.world * B * k / el3_max, 1),
                                'records_identical_to_timed_region': bool(recs3.tobytes() == recs.tobytes())}
        out['two_streams'] = {'steps': k, 'seconds': round(el2_max, 3), 'ms_per_step': round(el2_max / k * 1e3, 4),
                              'frames_per_s': round(env.world * B * k / el2_max, 1),
                              'records_identical_to_timed_region': bool(recs2.tobytes() == recs.tobytes())}
    if cpu_sample > 0:
        # rank 0 runs the CPU sample (and the parity gate) outside every timed region; the other ranks wait at the barrier
        if env.rank == 0:
            out['cpu'] = cpu_block(pfile, frames, recs, min(cpu_sample, B), P)
        env.barrier()
    return out


def fft_correlation_rate(P, crows, ccols, nimg=64):
    """Informational: what ONE stage of the reference's CPU path costs the way OpenCV does it.  cv2.matchTemplate computes
    TM_CCOEFF with a blocked float32 DFT, an order of magnitude fewer operations than the oracle's exact
    direct correlation; scipy.signal.fftconvolve of the same 250x250 image with the flipped 119x188 template, float32,
    one thread, is the closest stand-in available here (cv2 itself is absent).  Match stage only."""
    try:
        from scipy.signal import fftconvolve
    except Exception:
        return None
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(nimg, crows, ccols)).astype(np.float32)
    tpl = rng.integers(0, 256, size=(P.th, P.tw)).astype(np.float32)[::-1, ::-1].copy()
    fftconvolve(img[0], tpl, mode='valid')
    t0 = time.perf_counter()
    for i in range(nimg):
        fftconvolve(img[i], tpl, mode='valid')
    dt = time.perf_counter() - t0
    return {'value': round(nimg / dt, 1), 'unit': 'images/s', 'cores': 1,
            'what': 'scipy.signal.fftconvolve (float32, mode=valid) of %dx%d images with the %dx%d template: the match stage '
                    'alone, FFT-based as in OpenCV; not bit-exact, not the whole path' % (ccols, crows, P.tw, P.th)}


def record_matches_oracle(r, o, ndials):
    """The in-run parity gate's comparison of one GPU record with the oracle's: status, match position, the float32 match
    value BIT for bit, then the printed value (4 dials: the reference's '{:07.3f}' line) or the dial positions to 1e-9."""
    same = (int(r['status']) == o.status and int(r['match_x']) == o.match_x and int(r['match_y']) == o.match_y
            and np.float32(r['match_val']).tobytes() == np.float32(o.match_val).tobytes())
    if same and o.status == 0:
        if ndials == 4:
            same = '{:07.3f}'.format(float(r['value'])) == '{:07.3f}'.format(o.value)
        same = same and bool(np.allclose(r['pos'][:ndials], list(o.pos)[:ndials], rtol=0, atol=1e-9))
    elif same and o.status == 2:
        same = int(r['failed_dial']) == o.failed_dial
    elif same and o.status == 3:
        same = int(r['unreadable_mask']) == o.unreadable_mask
    return same


def cpu_block(pfile, frames, recs, S, P=None):
    """The CPU oracle (port of the reference's algorithm; the reference itself needs OpenCV 3.4.5, absent here) on
    the first S frames of batch 0: one thread (the reference is single-threaded) and all of this GPU's host cores.
    The same sample is the in-run parity gate."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as po
    sample = frames[:S].cpu().numpy()
    op = po.Params(pfile)
    po.process_frames(sample[:2], op)  # warm the library
    tc0 = time.perf_counter()
    ores = po.process_frames(sample, op)
    tc = time.perf_counter() - tc0
    ndials = int(P.ndials) if P is not None else 4
    mism = sum(0 if record_matches_oracle(recs[i], ores[i], ndials) else 1 for i in range(S))
    ncores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    ncores = min(ncores, 16, S)  # one GPU's share of the host
    parts = [sample[i::ncores] for i in range(ncores) if len(sample[i::ncores])]
    with ThreadPoolExecutor(max_workers=ncores) as pool:
        tm0 = time.perf_counter()
        list(pool.map(lambda part: po.process_frames(part, op), parts))
        tm = time.perf_counter() - tm0
    fft = None
    if P is not None:
        (H, W) = sample.shape[1:3]
        fft = fft_correlation_rate(P, min(P.rect_y1, H) - min(P.rect_y0, H), min(P.rect_x1, W) - min(P.rect_x0, W))
    return {'value': round(S / tc, 2), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'all_cores': {'value': round(S / tm, 2), 'unit': 'frames/s', 'cores': ncores},
            'note': 'the port computes TM_CCOEFF by exact direct integer correlation (positions x template MACs per frame) so that '
                    'it can be the bit-exact checker; the reference gets the same map from OpenCV\'s blocked float32 DFT, roughly an '
                    'order of magnitude fewer operations, so OpenCV on one core would be several times FASTER than this port. '
                    'fft_match_stage_only is that stage alone done the FFT way (informational).',
            'fft_match_stage_only': fft,
            'sample': 'first %d frames of batch 0 of the same workload through oracle/melf_oracle.c (exact direct '
                      'correlation, single thread, %.1f s; the reference itself needs OpenCV 3.4.5, absent here)' % (S, tc),
            'parity_mismatches_vs_gpu': mism,
            'parity_gate': 'status, match position, float32 match value bit for bit, printed value, dial positions to 1e-9'}


DIAG_LIB = os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_diag.so')


def stream_ceiling_child(env, FB, H, W, nbuf, steps):
    """fused_mask.stream_ceiling: run by a CHILD process that loads the diagnostic build (make -C meterelf_amd/csrc diag ->
    libmeterelf_hip_diag.so: the product's code + melf_stream_probe_dev and its bare-stream kernels) on this rank's GPU, while this
    process idles: bare streams and the fused kernel itself (same source, same launch shape as the product's) interleaved over the
    child's own buffers of the same size.  None when the diagnostic build is missing."""
    if not os.path.exists(DIAG_LIB):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), '--ceiling-child', '%d,%d,%d,%d,%d,%d' % (FB, H, W, nbuf, steps, env.dev_index)]
    envv = dict(os.environ, MELF_LIB_PATH=DIAG_LIB)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        envv.pop(k, None)
    try:
        p = subprocess.run(cmd, env=envv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
        return json.loads(lines[-1]) if p.returncode == 0 and lines else {'error': p.stderr.decode()[-400:]}
    except Exception as e:   # the ceiling is an annotation: the line goes out without it
        return {'error': repr(e)}


def ceiling_child_main(spec):
    """The child of stream_ceiling_child (MELF_LIB_PATH = the diagnostic build): prints the stream_ceiling object."""
    (FB, H, W, nbuf, steps, dev) = (int(x) for x in spec.split(','))
    import torch
    from meterelf_amd import _engine, _hip, _params
    assert hasattr(_hip.lib(), 'melf_stream_probe_dev'), 'the child needs the diagnostic build (MELF_LIB_PATH)'
    torch.cuda.set_device(dev)
    device = torch.device('cuda', dev)
    ctx = _hip.Context(_engine.make_blob(_params.load(os.path.join(GOLDEN, 'sample-images1', 'params.yml'))), dev)
    stream = torch.cuda.Stream(device=device).cuda_stream
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    frames = torch.randint(0, 256, (nbuf * FB, H, W, 3), dtype=torch.uint8, device=device, generator=g)
    masks = torch.empty((nbuf * FB, H, W), dtype=torch.uint8, device=device)
    alg_bytes = FB * H * W * 4

    def launch(i):
        b = i % nbuf
        ctx.hls_inrange_close_dev(frames.data_ptr() + b * FB * H * W * 3, FB, H, W, masks.data_ptr() + b * FB * H * W, stream=stream)
    tp0 = time.perf_counter()
    while time.perf_counter() - tp0 < 0.06:   # untimed preheat, as in fused_block: the clocks settle after 20-30 ms of load
        for i in range(16):
            launch(i)
        torch.cuda.synchronize()
    for i in range(max(6, nbuf)):
        launch(i)
    torch.cuda.synchronize()
    in_bytes = FB * H * W * 3
    variants = [('static', 0), ('static_prefetch', -1), ('queue2', 2), ('queue4', 4), ('comb', -2), ('comb_stride48', -3), ('comb_stride48_barriers', -4),
                ('comb_stride48_barriers_tables', -5)]
    ctx.set_profiling(1)
    ctx.timings()
    rounds = max(6, min(steps, 24))
    for i in range(rounds + 2):
        if i == 2:
            torch.cuda.synchronize()
            ctx.timings()       # the first two rounds are warm-up
        b = i % nbuf
        for (_name, ch) in variants:
            ctx.stream_probe_dev(frames.data_ptr() + b * in_bytes, in_bytes, masks.data_ptr() + b * FB * H * W, ch, stream=stream)
    torch.cuda.synchronize()
    (pms, pn) = ctx.timings()['k_stream_probe']
    # per variant: the launches alternate, so a second pass with one variant at a time gives the split
    per = {}
    for (name, ch) in variants:
        for i in range(rounds):
            b = i % nbuf
            ctx.stream_probe_dev(frames.data_ptr() + b * in_bytes, in_bytes, masks.data_ptr() + b * FB * H * W, ch, stream=stream)
        torch.cuda.synchronize()
        (vms, vn) = ctx.timings()['k_stream_probe']
        moved = (in_bytes // (48 * 1024)) * 64 * 1024
        per[name] = {'avg_launch_ms': round(vms / max(vn, 1), 4), 'GBps': round(moved / (vms / max(vn, 1) * 1e-3) / 1e9, 1)}
    # the kernel, right after, for a same-minute comparison
    for i in range(rounds):
        launch(i)
    torch.cuda.synchronize()
    (kms, kn) = ctx.timings()['k_fused_mask']
    ctx.set_profiling(0)
    best = max(v['GBps'] for v in per.values())
    k_gbs = alg_bytes / (kms / max(kn, 1) * 1e-3) / 1e9
    out = {'what': 'bare persistent 3:1 stream (48 B read + 16 B written per thread and step, lane-contiguous 16-byte loads, '
                   'non-temporal stores, 512 workgroups of 1024 threads, no arithmetic) over buffers of the same size in the same '
                   'rotation, dispatch time stamps like the kernel\'s, in a child process on the diagnostic build of the library; '
                   'static = grid-stride split, static_prefetch = the same with the next chunk requested before this one is stored '
                   '(the kernel\'s register prefetch), queueN = blocks of N chunks from a work queue',
           'variants': per, 'best_GBps': best, 'frac_of_hbm_peak': round(best / HBM_PEAK_GBS, 4),
           'interleaved_avg_launch_ms': round(pms / max(pn, 1), 4),
           'kernel_right_after': {'avg_launch_ms': round(kms / max(kn, 1), 4), 'GBps': round(k_gbs, 1),
                                  'frac_of_hbm_peak': round(k_gbs / HBM_PEAK_GBS, 4)},
           'kernel_frac_of_achievable': round(k_gbs / best, 4)}
    ctx.close()
    print(json.dumps(out))


def fused_block(env, ctx, FB, H, W, nbuf, steps, warmup, traffic, label, frames=None):
    """Fused HLS + inRange + closing over nbuf distinct input / output buffer pairs (the pairs together exceed the
    Infinity Cache several times over).  HBM roofline from the per-launch event times."""
    torch = env.torch
    g = torch.Generator(device=env.device)
    g.manual_seed(1234 + env.rank)
    if frames is None:
        frames = torch.randint(0, 256, (nbuf * FB, H, W, 3), dtype=torch.uint8, device=env.device, generator=g)
    masks = torch.empty((nbuf * FB, H, W), dtype=torch.uint8, device=env.device)

    def launch(i):
        b = i % nbuf
        ctx.hls_inrange_close_dev(frames.data_ptr() + b * FB * H * W * 3, FB, H, W, masks.data_ptr() + b * FB * H * W,
                                  stream=env.stream)
    # Untimed preheat, as the headline's 300 steps: after an idle stretch (this block's buffers have just been made) the chip needs
    # 20-30 ms of load before its clocks have settled -- launches 4-8 of a 1080p run are 15-20 % slower than launch 40
    # (tools/fused_first_launches.py, profiles/r06/fused_first_launches.txt) -- and W + K launches would sit inside that
    preheat = 0
    tp0 = time.perf_counter()
    while time.perf_counter() - tp0 < 0.06:
        for i in range(16):
            launch(preheat + i)
        env.sync()
        preheat += 16
    for i in range(max(warmup, nbuf)):
        launch(i)
    env.sync()
    ctx.set_profiling(1)
    ctx.timings()
    tf0 = time.perf_counter()
    for i in range(steps):
        launch(i)
    env.sync()
    tf = time.perf_counter() - tf0
    (fms, fn) = ctx.timings()['k_fused_mask']
    ctx.set_profiling(0)
    favg = fms / max(fn, 1)
    alg_bytes = FB * H * W * 4   # 3 B/px read + 1 B/px written
    gbs = alg_bytes / (favg * 1e-3) / 1e9
    (tr, src) = traffic.get(label + ':k_fused_mask')
    # the same launches alternating between two streams: one launch's ramp-up fills the other's tail
    two = None
    if nbuf >= 2:
        def launch2(i):
            b = i % nbuf
            ctx.hls_inrange_close_dev(frames.data_ptr() + b * FB * H * W * 3, FB, H, W, masks.data_ptr() + b * FB * H * W,
                                      stream=env.stream_objs[i % 2].cuda_stream)
        for i in range(4):
            launch2(i)
        env.sync()
        n2 = max(steps, 40)
        t20 = time.perf_counter()
        for i in range(n2):
            launch2(i)
        env.sync()
        t2 = time.perf_counter() - t20
        two = {'launches': n2, 'ms_per_launch': round(t2 / n2 * 1e3, 4), 'GBps': round(alg_bytes * n2 / t2 / 1e9, 1),
               'frac_of_hbm_peak': round(alg_bytes * n2 / t2 / 1e9 / HBM_PEAK_GBS, 4)}
    # What the part streams for this traffic mix: a bare persistent stream of the same mix and launch shape, measured in this run by
    # a child process on the DIAGNOSTIC build of the library (the product library does not carry measurement kernels)
    # (single-GPU runs only: a multi-rank job does not start further GPU processes beside its ranks)
    ceiling = stream_ceiling_child(env, FB, H, W, nbuf, steps) if (env.rank == 0 and env.world == 1) else None
    del masks
    return {
        'workload': 'B=%d %dx%d uniform-random u8 frames, fused HLS+inRange+closing only, %d distinct buffer pairs '
                    '(%.2f GB) in rotation' % (FB, W, H, nbuf, nbuf * alg_bytes / 1e9),
        'frames_per_s': round(FB * steps / tf, 1),
        'untimed_preheat_launches': preheat,
        'roofline': {'kernel': 'k_fused_mask', 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': tr, 'traffic_source': src,
                     'avg_launch_ms': round(favg, 4), 'launches': fn,
                     'algorithmic': '%d B/frame x %d frames/launch' % (H * W * 4, FB),
                     # the kernel needs the vector units as well (12-15 instructions per pixel): their busy share from the PMC pass
                     'vector_units_busy': (round(traffic.valu_entry(label + ':k_fused_mask')['valu_busy_frac'], 4)
                                           if traffic.valu_entry(label + ':k_fused_mask') else None)},
        'stream_ceiling': ceiling,
        'two_streams': two,
    }


def config5_block(env, cfg3_frames, steps, warmup, traffic):
    """BASELINE config 5, one GPU's share: 512 frames of 1920x1080, meter_rect inside the frame, six dials.
    (a) the fused per-pixel stage over whole frames (the HBM stress), (b) the full path (match on the 250x250
    meter_rect + six dials)."""
    from meterelf_amd import _hip
    torch = env.torch
    d = config5_params_dir() if env.rank == 0 else None
    try:
        (ctx, names) = env.make_context(os.path.join(d, 'params.yml') if d else None)
        oparams = None
        if d and 'cpu' in env.blocks:
            from oracle import pyoracle as po
            oparams = po.Params(os.path.join(d, 'params.yml'))
            oparams.load_template()   # while the directory exists
    finally:
        if d:
            shutil.rmtree(d, ignore_errors=True)
    (B5, H5, W5) = (env.args.batch5, 1080, 1920)
    g = torch.Generator(device=env.device)
    g.manual_seed(1080 + env.rank)
    frames = torch.randint(0, 256, (B5, H5, W5, 3), dtype=torch.uint8, device=env.device, generator=g)
    # the meter: config 3's synthesised frames' meter_rect crops pasted at the 1080p meter_rect
    k = cfg3_frames.shape[0]
    for i0 in range(0, B5, k):
        m = min(k, B5 - i0)
        frames[i0:i0 + m, 420:670, 1210:1460] = cfg3_frames[:m, 160:410, 50:300]
    env.sync()
    fused = fused_block(env, ctx, B5, H5, W5, 1, max(12, steps), 6, traffic, 'config5', frames=frames)
    fused['workload'] = ('B=%d frames of 1920x1080 (%.2f GB in + %.2f GB out per launch), fused HLS+inRange+closing over '
                         'whole frames' % (B5, B5 * H5 * W5 * 3 / 1e9, B5 * H5 * W5 / 1e9))
    d_results = torch.zeros(2 * B5 * _hip.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=env.device)   # two record slices: one per caller stream
    # untimed preheat (the context and its buffers have just been made: the chip's clocks settle 20-30 ms after the load begins)
    # (this rank's own calls and syncs only: a loop bounded by TIME must not hold a barrier -- the ranks leave it after different counts)
    (tp0, pre5) = (time.perf_counter(), 0)
    while time.perf_counter() - tp0 < 0.06:
        for _ in range(16):
            ctx.process_batch_dev(frames.data_ptr(), B5, H5, W5, d_results_ptr=d_results.data_ptr(), want_host=False, stream=env.stream)
        env.sync()
        pre5 += 16
    timed_steps(env, ctx, frames, B5, 1, H5, W5, d_results, max(warmup, 1))
    ctx.set_profiling(1)
    ctx.timings()
    timed_steps(env, ctx, frames, B5, 1, H5, W5, d_results, 3)
    kt_all = ctx.timings()
    ctx.set_profiling(2)
    (el1, recs1) = timed_steps(env, ctx, frames, B5, 1, H5, W5, d_results, steps)
    kt = ctx.timings()
    ctx.set_profiling(0)
    (el1_max, _p) = max_over_ranks(env, el1)
    # the timed figure: like the headline, the same calls alternating between two caller streams (two lanes), no stamps
    if env.args.no_resident_hint:
        (el, recs) = (el1, recs1)
    else:
        ns5 = 1 if env.args.mode == 'resident' else 2
        ctx.set_frames_resident(ns5 == 1)
        timed_steps(env, ctx, frames, B5, 1, H5, W5, d_results, max(warmup, 4), nstreams=ns5)
        (el, recs) = timed_steps(env, ctx, frames, B5, 1, H5, W5, d_results, steps, nstreams=ns5)
        ctx.set_frames_resident(False)
    (recs, recs1) = (recs[:B5], recs1[:B5])
    (el_max, _p) = max_over_ranks(env, el)
    P = ctx.params
    out = {'workload': 'B=%d per GPU, 1920x1080 frames (%d B/frame, %.2f GB per GPU), 6 dials, meter_rect 250x250 inside '
                       'the frame' % (B5, H5 * W5 * 3, B5 * H5 * W5 * 3 / 1e9),
           'fused_mask': fused,
           'full_path': {'frames_per_s': round(env.world * B5 * steps / el_max, 1), 'ms_per_step': round(el_max / steps * 1e3, 4),
                         'single_lane': {'ms_per_step': round(el1_max / steps * 1e3, 4), 'frames_per_s': round(env.world * B5 * steps / el1_max, 1),
                                         'records_identical_to_timed_region': bool(recs1.tobytes() == recs.tobytes()),
                                         'what': 'the same steps as melf_process_batch_dev calls on one stream: the launches roofline / kernel_ms describe'},
                         'dials': int(P.ndials), 'frames_read_ok': int((recs['status'] == 0).sum()), 'untimed_preheat_steps': pre5,
                         'kernel_ms': {k_: round(ms / n, 4) for (k_, (ms, n)) in kt_all.items() if n},
                         'match_layout': ctx.last_match(),
                         'roofline': match_roofline(P, H5, W5, kt, B5 * steps, (None, None))}}
    # in-run parity gate for THIS launch shape (512 resident 1080p frames, 6 dials, one call): 64 frames spread over the batch
    # through the oracle, rank 0, outside every timed region
    if oparams is not None:
        from oracle import pyoracle as po
        pick = np.unique(np.linspace(0, B5 - 1, min(64, B5)).astype(np.int64))
        sample = frames[torch.from_numpy(pick).to(env.device)].cpu().numpy()
        ores = po.process_frames(sample, oparams)
        mism = sum(0 if record_matches_oracle(recs[i], ores[k], int(P.ndials)) else 1 for (k, i) in enumerate(pick.tolist()))
        out['full_path']['parity_gate'] = {'oracle_frames': int(len(pick)), 'parity_mismatches_vs_gpu': mism,
                                           'what': 'frames spread evenly over the batch of the timed launch: status, match position, '
                                                   'float32 match value bit for bit, six dial positions to 1e-9'}
    env.barrier()
    ctx.close()
    return out


def hostfed_block(env, ctx, frames, B, H, W):
    """melf_process_batch: frames in pageable HOST memory in, records out (PCIe inclusive; never `value`)."""
    host = frames[:B].cpu().numpy()
    ctx.process_batch(host)   # warm-up at full size: the pinned staging buffers are allocated on first use
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        recs = ctx.process_batch(host)
    dt = (time.perf_counter() - t0) / reps
    P = ctx.params
    crop_bytes = (min(P.rect_y1, H) - min(P.rect_y0, H)) * (min(P.rect_x1, W) - min(P.rect_x0, W)) * 3
    return {'workload': '%d frames %dx%d in pageable host memory -> records on the host, one melf_process_batch call' % (B, W, H),
            'frames_per_s': round(B / dt, 1), 'ms_per_call': round(dt * 1e3, 3),
            'pcie_GBps_crop_bytes': round(B * crop_bytes / dt / 1e9, 2),
            'frame_GBps_equivalent': round(B * H * W * 3 / dt / 1e9, 2),
            'frames_read_ok': int((recs['status'] == 0).sum())}


def jpeg_block(ctx, sample_dir, H, W, traffic=None):
    from meterelf_amd import _hip
    from meterelf_amd._image import imread_bgr
    jfiles = [f for f in sorted(glob.glob(os.path.join(GOLDEN, sample_dir, '*.jpg'))) if os.path.basename(f) not in REJECTED]
    blobs = [open(f, 'rb').read() for f in jfiles]
    keep = [i for (i, b) in enumerate(blobs) if _hip.jpeg_probe(b)[:3] == (H, W, True)]
    (jfiles, blobs) = ([jfiles[i] for i in keep], [blobs[i] for i in keep])
    if not blobs:
        return None
    JB = 1024
    batch = [blobs[i % len(blobs)] for i in range(JB)]
    (jf, jst) = ctx.jpeg_decode(blobs[:8], H, W)
    same = bool((jst == 0).all()) and all(np.array_equal(jf[i], imread_bgr(f)) for (i, f) in enumerate(jfiles[:8]))
    table = _hip.file_table(batch)       # pointers and sizes marshalled once: the timed calls are the C entry point's
    ctx.jpeg_process_batch(batch, H, W, table)  # warm-up (allocations)
    ctx.jpeg_process_batch(batch, H, W, table)
    reps = 20
    tj0 = time.perf_counter()
    for _ in range(reps):
        (jrecs, jstatus) = ctx.jpeg_process_batch(batch, H, W, table)
    tj = (time.perf_counter() - tj0) / reps
    # per-kernel times from a separate pass: event records around every kernel of every chunk keep the chunks' kernels
    # from overlapping, so the timed calls above run without them
    ctx.set_profiling(1)
    ctx.timings()
    ctx.jpeg_process_batch(batch, H, W)
    jt = ctx.timings()
    ctx.set_profiling(0)
    # the reference's own entry point on file NAMES (meterelf/_api.py:9-33): get_meter_values reads, decodes and reads out
    # 1024-file chunks inside the library (two chunks in flight) and turns the records into MeterImageData objects
    from meterelf_amd import get_meter_values, release_cached_contexts
    pfile = os.path.join(GOLDEN, sample_dir, 'params.yml')
    names = [jfiles[i % len(jfiles)] for i in range(64 * 1024)]  # a long list: the per-call costs (params, calibration blob) stop mattering
    dev = os.environ.get('METERELF_DEVICES')
    os.environ['METERELF_DEVICES'] = '%d' % ctx.device   # ONE context on this rank's GPU, whatever the caller's environment says
    from meterelf_amd import _api

    def host_block(stats, seconds):
        """Per-chunk host times of a get_meter_values run: the library's stages (read / turn / enqueue / GPU wait, per device
        pipeline) and the Python side (begin, blocked in end, record conversion), plus what the pools were sized from."""
        chunks = max(stats['chunks'], 1)
        lib = stats['library']
        out = {'os_cpu_count': os.cpu_count(), 'affinity_cores': len(os.sched_getaffinity(0)), 'chunks': stats['chunks'],
               'python_ms_per_chunk': {'begin_marshal': round(stats['s_begin'] / chunks * 1e3, 3), 'blocked_in_end': round(stats['s_end_wait'] / chunks * 1e3, 3),
                                       'records_to_objects': round(stats['s_convert'] / chunks * 1e3, 3)},
               'wall_ms_per_chunk': round(seconds / chunks * 1e3, 3)}
        if lib:
            calls = max(sum(x['calls'] for x in lib), 1.0)
            out['library_ms_per_chunk'] = {k[3:]: round(sum(x[k] for x in lib) / calls, 3) for k in ('ms_read', 'ms_turn_wait', 'ms_enqueue', 'ms_gpu_wait')}
            out['io_threads'] = int(lib[0]['io_threads'])
            out['host_threads'] = int(lib[0]['host_threads'])
            out['pool_cores'] = int(lib[0]['cores'])
            out['devices_in_process'] = int(lib[0]['devices_in_process'])
            out['pipelines'] = len(lib)
        return out

    # what the file system allows the read stage: open() + close() alone, on the library's I/O pool (best of 5)
    probe = min(_hip.files_open_probe(names[:1024], ctx.device) for _ in range(5))
    open_probe = {'ms_per_1024_files': round(probe[0], 3), 'threads': probe[1],
                  'what': 'open() + close() of 1024 of the files on the I/O pool, nothing read: the file system\'s floor under the read stage'}
    sum(1 for _ in get_meter_values(pfile, names[:2048]))  # warm-up: the context the API keeps between calls
    _api.api_stats(reset=True)
    tg0 = time.perf_counter()
    n_api = sum(1 for r in get_meter_values(pfile, names) if r.error is None)
    tg = time.perf_counter() - tg0
    host1 = host_block(_api.api_stats(reset=True), tg)
    host1['open_close_probe'] = open_probe
    try:
        with open('/sys/fs/cgroup/cpu.max') as fp:
            host1['cgroup_cpu_max'] = fp.read().strip()
    except OSError:
        pass
    release_cached_contexts()
    # the same over TWO contexts on this GPU (METERELF_DEVICES=0,0: the multi-device fan-out of the API -- one reader, one host
    # thread, one begin / end pipeline per entry -- exercised on the one GPU a rank has; on a node the entries are the node's GPUs)
    os.environ['METERELF_DEVICES'] = '%d,%d' % (ctx.device, ctx.device)
    try:
        sum(1 for _ in get_meter_values(pfile, names[:4096]))
        _api.api_stats(reset=True)
        tg20 = time.perf_counter()
        n_api2 = sum(1 for r in get_meter_values(pfile, names) if r.error is None)
        tg2 = time.perf_counter() - tg20
        host2 = host_block(_api.api_stats(reset=True), tg2)
    finally:
        if dev is None:
            os.environ.pop('METERELF_DEVICES', None)
        else:
            os.environ['METERELF_DEVICES'] = dev
        release_cached_contexts()
    return {'workload': '%d JPEG files (%d distinct %s fixtures, %.1f KB average) -> decode + full reading path, '
                        'file bytes in host memory to result records (melf_jpeg_process_batch, mean of 20 calls; the pointer table built once)' % (JB, len(blobs), sample_dir, sum(map(len, blobs)) / len(blobs) / 1024),
            'files_per_s': round(JB / tj, 1), 'ms_per_call': round(tj * 1e3, 3),
            'get_meter_values': {'files_per_s': round(len(names) / tg, 1), 'files': len(names), 'values_read': n_api, 'devices': 1, 'host': host1,
                                 'what': 'meterelf_amd.get_meter_values(params.yml, file names): the reference API, files read '
                                         'from the page cache inside the library, MeterImageData objects out',
                                 'two_contexts_on_this_gpu': {'files_per_s': round(len(names) / tg2, 1), 'values_read': n_api2, 'host': host2,
                                                              'what': 'METERELF_DEVICES=d,d: the API\'s multi-device fan-out with both entries on this GPU '
                                                                      '(same host cores, same GPU: a functional figure, not a scaling one)'}},
            'kernel_ms_per_call': {k: round(ms, 4) for (k, (ms, c)) in jt.items() if c and k.startswith('k_jpeg')},
            'kernel_launches_per_call': {k: c for (k, (ms, c)) in jt.items() if c and k.startswith('k_jpeg')},
            'roofline_jpeg': jpeg_roofline(jt, traffic),
            'decoded_frames_equal_libjpeg_turbo': same, 'files_ok': int((jstatus == 0).sum())}


def dry_run(args):
    """Launch / rendezvous / collective plumbing without a GPU (CI): no step is run, nothing is measured."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    ranks = 1
    if 'RANK' in os.environ:
        dist.init_process_group(backend='gloo')
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        ranks = int(t.item())
        from meterelf_amd import _dist, _engine, _params
        (blob, names) = (None, None)
        if rank == 0:
            params = _params.load(os.path.join(GOLDEN, args.sample_dir, 'params.yml'))
            (blob, names) = (_engine.make_blob(params), list(params.dial_names))
        (blob, _dev, names) = _dist.broadcast_blob(blob, names, src=0)
        digest = hashlib.sha256(blob.tobytes()).hexdigest()[:16]
        allsums = [None] * world
        dist.all_gather_object(allsums, digest)
        assert len(set(allsums)) == 1, allsums
        dist.barrier()
    if rank == 0:
        print(json.dumps({'dry_run': True, 'metric': 'none (launch rehearsal, nothing measured)', 'value': None, 'n_gpus': world,
                          'collective_ranks': ranks, 'backend': 'gloo' if 'RANK' in os.environ else None}))
        sys.stdout.flush()
    if 'RANK' in os.environ:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed steps (default 200 = 60 ms of GPU time: one 1 ms hiccup of the host or the '
                    'profiler inside a 20-step region is a 17 %% error)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='frames per GPU per step')
    ap.add_argument('--nbuf', type=int, default=4, help='distinct batches the steps rotate over')
    ap.add_argument('--batch5', type=int, default=512, help='config 5: 1080p frames per GPU')
    ap.add_argument('--sample-dir', default='sample-images1')
    ap.add_argument('--cpu-sample', type=int, default=768, help='frames timed through the CPU oracle (about 10 s on one core)')
    ap.add_argument('--sustained', type=float, default=2.0, help='seconds of back-to-back steps in the sustained block')
    ap.add_argument('--streams', type=int, default=1, help='caller streams the steps alternate between (2: steps overlap on the context\'s two lanes)')
    ap.add_argument('--no-resident-hint', action='store_true',
                    help='time the headline, config 4 and config 5 on ONE caller stream (every kernel behind the previous one) instead of '
                         'two; the single_lane objects carry that figure either way')
    ap.add_argument('--mode', default='two_streams', choices=['two_streams', 'resident'], help='how the timed regions drive the two lanes: consecutive steps on two caller '
                    'streams, or one caller stream with melf_ctx_set_frames_resident')
    ap.add_argument('--preheat', type=int, default=300, help='untimed steps run immediately before the timed region (clock settling)')
    ap.add_argument('--skip', default='', help='comma list of blocks to skip: ' + ','.join(ALL_BLOCKS))
    ap.add_argument('--only', default='', help='comma list of extra blocks to run (default: all)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend (nccl = RCCL)')
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal: ranks share the visible GPUs (gloo backend only)')
    ap.add_argument('--dry-run', action='store_true', help='launcher / collective rehearsal without a GPU; measures nothing')
    ap.add_argument('--ceiling-child', default='', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.ceiling_child:
        return ceiling_child_main(args.ceiling_child)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.dry_run:
        return dry_run(args)

    blocks = set(ALL_BLOCKS)
    if args.only:
        blocks = set(x for x in args.only.split(',') if x)
    blocks -= set(x for x in args.skip.split(',') if x)

    env = Env(args)
    env.blocks = blocks
    torch = env.torch
    (rank, world) = (env.rank, env.world)
    rccl_ranks = None
    if env.dist is not None:
        one = torch.ones(1, dtype=torch.int32, device=env.device if env.backend == 'nccl' else 'cpu')
        env.dist.all_reduce(one)
        rccl_ranks = int(one.item()) if env.backend == 'nccl' else None
    traffic = Traffic()
    single = world == 1

    # ---- headline: BASELINE config 3 ----
    pfile = os.path.join(GOLDEN, args.sample_dir, 'params.yml')
    main_label = 'config3' if args.sample_dir == 'sample-images1' else 'config4'
    full = full_path_block(env, pfile, args.sample_dir, 2024, args.steps, args.warmup, args.batch, args.nbuf,
                           args.sustained if 'sustained' in blocks else 0.0, traffic,
                           (args.cpu_sample if single else min(args.cpu_sample, 256)) if 'cpu' in blocks else 0, main_label,
                           two_stream_s=1.0 if 'twostream' in blocks else 0.0)
    (ctx, P, H, W, B) = (full['ctx'], full['P'], full['H'], full['W'], args.batch)
    n_ok = int((full['recs'][:B]['status'] == 0).sum())
    roofline = full['roofline']
    roofline['k_dials_avg_launch_ms'] = full['kernel_ms'].get('k_dials')
    roofline['k_prep_avg_launch_ms'] = full['kernel_ms'].get('k_lplane')
    roofline_dials = dials_roofline(full['kernel_ms'], traffic, main_label)
    roofline_prep = prep_roofline(full['P'], full['H'], full['W'], full['kernel_ms'], traffic, main_label, args.batch)

    fused = None
    if 'fused' in blocks:
        fused = fused_block(env, ctx, 256, 640, 480, 4, args.steps, args.warmup, traffic, 'config2')
    # host-side blocks: rank 0 alone (they use the host's cores), the other ranks wait at the barrier
    (hostfed, jpeg) = (None, None)
    if rank == 0:
        hostfed = hostfed_block(env, ctx, full['frames'], B, H, W) if 'hostfed' in blocks else None
        jpeg = jpeg_block(ctx, args.sample_dir, H, W, traffic) if 'jpeg' in blocks else None
    env.barrier()

    cfg5 = None
    if 'config5' in blocks and args.sample_dir == 'sample-images1':
        cfg5 = config5_block(env, full['frames'], args.steps, args.warmup, traffic)
    cfg3_frames = full.pop('frames')
    del cfg3_frames
    ctx.close()

    cfg4 = None
    if 'config4' in blocks and args.sample_dir == 'sample-images1':
        torch.cuda.empty_cache()
        p4 = os.path.join(GOLDEN, 'sample-images2', 'params.yml')
        f4 = full_path_block(env, p4, 'sample-images2', 2025, args.steps, args.warmup, args.batch, args.nbuf, 0.0, traffic,
                             min(args.cpu_sample, 256) if 'cpu' in blocks else 0, 'config4',
                             two_stream_s=1.0 if 'twostream' in blocks else 0.0)
        cfg4 = {'workload': 'Batch=%d per GPU (%d in total), sample-images2 params (crop 135x220, 561 match positions), '
                            'calibration blob broadcast from rank 0%s, %d distinct batches in rotation'
                            % (B, B * world, ' over RCCL' if env.backend == 'nccl' else '', args.nbuf),
                'timed_region': f4['mode'] + ', like the headline; roofline and kernel_ms describe the undisturbed launches of single_lane',
                'frames_per_s': round(world * B * args.steps / f4['elapsed'], 1), 'ms_per_step': round(f4['elapsed'] / args.steps * 1e3, 4),
                'per_rank_ms_per_step': f4['per_rank_ms'], 'frames_read_ok_batch0': int((f4['recs'][:B]['status'] == 0).sum()),
                'kernel_ms': f4['kernel_ms'], 'step_events': f4.get('step_events'), 'match_layout': f4.get('match_layout'),
                'roofline': f4['roofline'], 'roofline_dials': dials_roofline(f4['kernel_ms'], traffic, 'config4'),
                'roofline_prep': prep_roofline(f4['P'], f4['H'], f4['W'], f4['kernel_ms'], traffic, 'config4', args.batch),
                'two_streams': f4.get('two_streams'), 'single_lane': f4.get('single_lane'), 'resident_hint': f4.get('resident_hint'),
                'cpu_baseline': f4.get('cpu')}
        f4['ctx'].close()

    if rank == 0:
        elapsed = full['elapsed']
        line = {
            'metric': 'frames/sec (640x480), full pipeline, digits identical to the oracle',
            'value': round(world * B * args.steps / elapsed, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': 'Batch=%d per GPU, full pipeline incl. TM_CCOEFF match + needle reading, 4 dials/frame, '
                                   '%s params, frames %dx%d synthesised from the readable fixtures (shift +-8, noise sigma 2), '
                                   '%d distinct batches in rotation (%.2f GB of frames per GPU)'
                                   % (B, args.sample_dir, W, H, args.nbuf, args.nbuf * B * H * W * 3 / 1e9),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'frames_read_ok_batch0': n_ok,
                       'untimed_preheat_steps': args.preheat, 'mode': full['mode']},
            'per_rank_ms_per_step': full['per_rank_ms'], 'rccl_ranks': rccl_ranks, 'backend': env.backend,
            'kernel_ms': full['kernel_ms'], 'step_events': full.get('step_events'), 'match_layout': full.get('match_layout'),
            'roofline': roofline, 'roofline_dials': roofline_dials, 'roofline_prep': roofline_prep, 'sustained': full.get('sustained'), 'two_streams': full.get('two_streams'),
            'single_lane': full.get('single_lane'), 'resident_hint': full.get('resident_hint'),
            'cpu_baseline': full.get('cpu'),
            'fused_mask': fused, 'config4': cfg4, 'config5': cfg5, 'host_fed': hostfed, 'jpeg_decode': jpeg,
        }
        print(json.dumps(line))
        sys.stdout.flush()
    if env.dist is not None:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == '__main__':
    main()
