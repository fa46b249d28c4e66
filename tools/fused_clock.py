#!/usr/bin/env python3
"""Where a k_fused_mask_lut launch spends its time (diagnostic build `make -C meterelf_amd/csrc stamp`): per workgroup, the
100 MHz real-time clock at its start, when its tables are in LDS, when its first pass has been stored, at its end -- and what
wave priorities change about it (MELF_FUSED_PRIO, an experiment of the diagnostic build only: 0 none, 2 the CU's second
workgroup favoured, 5..9 each of the two favoured half the time by the clock).
    python3 tools/fused_clock.py [HxW] [batch]      FUSED_PRIO_MODES=0,2,5,7,9"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_stamp.so'))
import numpy as np
import torch

from meterelf_amd import _engine, _hip, _params

(H, W) = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '480x640').split('x'))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev)
g.manual_seed(1234)
NB = 4
frames = torch.randint(0, 256, (NB * B, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
masks = torch.empty((NB * B, H, W), dtype=torch.uint8, device=dev)
L = _hip.lib()
for i in range(12):
    b = i % NB
    ctx.hls_inrange_close_dev(frames.data_ptr() + b * B * H * W * 3, B, H, W, masks.data_ptr() + b * B * H * W, stream=stream)
torch.cuda.synchronize()
MODES = [int(v) for v in os.environ.get('FUSED_PRIO_MODES', '0,2,5,7,9').split(',')]
for rep in range(2 * len(MODES)):
    os.environ['MELF_FUSED_PRIO'] = str(MODES[rep // 2])
    b = rep % NB
    ctx.hls_inrange_close_dev(frames.data_ptr() + b * B * H * W * 3, B, H, W, masks.data_ptr() + b * B * H * W, stream=stream)
    torch.cuda.synchronize()
    nwg = 512
    st = np.zeros((nwg, 8), np.uint64)
    assert L.melf_debug_fused_stamps(st.ctypes.data_as(C.c_void_p), nwg) == 0
    st = st[st[:, 3] > 0]
    t = st[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    (start, tab, first, end) = (us[:, 0], us[:, 1], us[:, 2], us[:, 3])
    hw = st[:, 5].astype(np.int64)
    (cu, sh, se) = ((hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7)   # gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    xcc = st[:, 4].astype(int)
    place = xcc * 1000 + se * 100 + sh * 50 + cu
    pairs = {}
    for (k, pl) in enumerate(place):
        pairs.setdefault(int(pl), []).append(k)
    idx = np.arange(len(end))
    (old, young) = (end[idx < 256], end[idx >= 256])
    bytes_total = B * H * W * 4
    print('prio mode %s: start of the last workgroup %.1f us | tables in LDS median %.1f | first pass stored median %.1f max %.1f | '
          'ends: first workgroup of a CU mean %.1f, second %.1f, all: min %.1f median %.1f max %.1f us | %d CUs x %s workgroups | %.2f TB/s over the span'
          % (os.environ['MELF_FUSED_PRIO'], start.max(), np.median(tab), np.median(first), first.max(), old.mean(), young.mean(), end.min(), np.median(end),
             end.max(), len(pairs), sorted(set(len(v) for v in pairs.values())), bytes_total / end.max() / 1e6), flush=True)
ctx.close()
