#!/usr/bin/env python3
"""Soak of the fused-mask launch's work queue (diagnostic build): random (frames, rows, columns) cases, the masks of the default
launch (small segments from the queue wherever launch_lut_t's rule says so) against the static split of the same frames
(MELF_FUSED_DYN=0), whole arrays compared; concurrent launches on two streams share the ring of queue slots.
    python3 tools/soak_fused_queue.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_diag.so'))
import torch

from meterelf_amd import _engine, _hip, _params

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = _hip.Context(_engine.make_blob(_params.load(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml'))), 0)
dev = torch.device('cuda', 0)
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream()
g = torch.Generator(device=dev)
g.manual_seed(99)
rng = __import__('random').Random(5)
t0 = time.time()
(cases, queued, px) = (0, 0, 0)
while time.time() - t0 < budget:
    W = 16 * rng.randint(6, 130)
    H = rng.randint(40, 1300)
    n = rng.randint(8, max(8, min(600, int(1.2e9 / (H * W * 3)))))
    frames = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
    # needle-coloured patches so that the masks are not empty
    frames[:, ::3, :, 2] = 200
    frames[:, ::3, :, 1] = 30
    frames[:, ::3, :, 0] = 40
    (m_auto, m_static, m_two) = (torch.empty((n, H, W), dtype=torch.uint8, device=dev) for _ in range(3))
    os.environ.pop('MELF_FUSED_DYN', None)
    ctx.hls_inrange_close_dev(frames.data_ptr(), n, H, W, m_auto.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    # two launches in flight on two streams (halves of the batch; the second half may or may not take the queue)
    torch.cuda.synchronize()
    h = n // 2
    ctx.hls_inrange_close_dev(frames.data_ptr(), h, H, W, m_two.data_ptr(), stream=s1.cuda_stream)
    ctx.hls_inrange_close_dev(frames.data_ptr() + h * H * W * 3, n - h, H, W, m_two.data_ptr() + h * H * W, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    os.environ['MELF_FUSED_DYN'] = '0'
    ctx.hls_inrange_close_dev(frames.data_ptr(), n, H, W, m_static.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(m_auto, m_static), ('queue vs static', n, H, W)
    assert torch.equal(m_two, m_static), ('two streams vs static', n, H, W)
    assert int((m_static > 0).sum()) > 0
    # launch_lut_t's rule, restated (k_hls.hip): does this launch take the queue?
    G16 = W >> 4
    RC = min(1024 // G16, 256)
    P = max(2, int((65536.0 / (W * 3) + 4.0) / RC + 0.999))
    while P * RC - 4 < 16:
        P += 1
    own = P * RC - 4
    segs = (512 + n - 1) // n
    seg_rows = (H + segs - 1) // segs
    if seg_rows < 32:
        seg_rows = min(H, 32)
    segs = (H + seg_rows - 1) // seg_rows
    st_total = n * segs
    st_rows = -(-st_total // min(st_total, 512)) * seg_rows
    q_rows = n * H * (1.0 + 4.0 / own) / 512.0 + own
    dr0 = min(own, H)
    ds = -(-H // dr0)
    queued += 1 if ((n * H / 512.0 >= 6.0 * own or q_rows < 0.9 * st_rows) and n * ds > 512) else 0
    cases += 1
    px += n * H * W
    del frames, m_auto, m_static, m_two
# many queue launches outstanding on three streams at once (a slot of the queue ring belongs to one stream of one context)
(H, W, n) = (1080, 1920, 64)
base = torch.randint(0, 256, (n + 32, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
base[:, ::3, :, 2] = 200
base[:, ::3, :, 1] = 30
base[:, ::3, :, 0] = 40
streams = [torch.cuda.Stream() for _ in range(3)]
outs = [[torch.empty((n, H, W), dtype=torch.uint8, device=dev) for _ in range(4)] for _ in streams]
os.environ.pop('MELF_FUSED_DYN', None)
torch.cuda.synchronize()
LAUNCHES = 40
for j in range(LAUNCHES):
    for (si, st) in enumerate(streams):
        off = (7 * j + 11 * si) % 32
        ctx.hls_inrange_close_dev(base.data_ptr() + off * H * W * 3, n, H, W, outs[si][j % 4].data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
os.environ['MELF_FUSED_DYN'] = '0'
chk = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
for (si, st) in enumerate(streams):
    for j in range(LAUNCHES - 4, LAUNCHES):
        off = (7 * j + 11 * si) % 32
        ctx.hls_inrange_close_dev(base.data_ptr() + off * H * W * 3, n, H, W, chk.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(chk, outs[si][j % 4]), ('outstanding launches', si, j)
print('%d queue launches of %d x 1080p outstanding on 3 streams: the last 4 of each stream identical to the static split' % (3 * LAUNCHES, n))
print('%d cases (%d of them long enough for the queue), %.1f G pixels, masks identical to the static split in all; %.0f s' % (cases, queued, px / 1e9, time.time() - t0))
