#!/usr/bin/env python3
"""Phase times of k_dials waves from in-kernel stamps (diagnostic build `make -C meterelf_amd/csrc stamp`)."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_stamp.so'))
import numpy as np
import torch

import bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr

sd = sys.argv[1] if len(sys.argv) > 1 else 'sample-images1'
pfile = os.path.join(ROOT, 'tests', 'golden', sd, 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', sd, '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
# optional third argument: which of the four batches tools/run_stage.py and bench.py rotate over (they differ: one slow wave makes a slow launch)
KB = int(sys.argv[3]) if len(sys.argv) > 3 else 0
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), (KB + 1) * B, 2024, dev)[KB * B:]
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    ctx.process_batch_dev(frames.data_ptr(), B, H, W, want_host=False, stream=stream)
torch.cuda.synchronize()
n = 4 * B
buf = np.zeros((n, 8), np.uint64)
assert _hip.lib().melf_debug_dials_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf[:, :6].astype(np.float64)
names = ['partials min/max + sync', 'dial colour (5x5 mean)', 'window pixels -> in-range mask', 'closing, flood, labelling, areas', 'momentum + angle passes']
tot = t[:, 5] - t[:, 0]
print('waves %d, total cycles per wave: median %.0f p90 %.0f' % (n, np.median(tot), np.sort(tot)[n * 9 // 10]))
t8 = buf.astype(np.float64)
for (label, a, b) in (('  pixel phase: candidate prefilter', 2, 6), ('  pixel phase: scan + list', 6, 7), ('  pixel phase: exact test of candidates', 7, 3)):
    dlt = t8[:, b] - t8[:, a]
    print('  %-42s median %8.0f cycles  p90 %8.0f' % (label, np.median(dlt), np.sort(dlt)[n * 9 // 10]))
for k in range(5):
    dlt = t[:, k + 1] - t[:, k]
    print('  %-34s median %8.0f cycles (%4.1f %%)  p90 %8.0f' % (names[k], np.median(dlt), 100 * np.median(dlt) / np.median(tot), np.sort(dlt)[n * 9 // 10]))
# per dial (wave w of a workgroup = dial w): which dials are the slow ones
nd = int(ctx.params.ndials)
for d in range(nd):
    sel = np.arange(n) % nd == d
    ex = t8[sel, 3] - t8[sel, 7]
    pf = t8[sel, 6] - t8[sel, 2]
    print('  dial %d: total median %6.0f p90 %6.0f | prefilter median %6.0f | exact test median %6.0f p90 %6.0f max %6.0f' % (
        d, np.median(tot[sel]), np.sort(tot[sel])[sel.sum() * 9 // 10], np.median(pf), np.median(ex), np.sort(ex)[sel.sum() * 9 // 10], ex.max()))
# time line (round 5): where the launch's last waves lost their time -- the 100 MHz real-time counter at the phase boundaries
# (the shader-clock counters of different CUs cannot be compared) and every wave's CU / SIMD (HW_ID)
real = np.zeros((n, 8), np.uint64)
assert _hip.lib().melf_debug_dials_real(real.ctypes.data_as(C.c_void_p), n) == 0
r = real[:, :6].astype(np.float64) * 0.01   # us
r -= r[:, 0].min()
hw = (real[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
cands = (real[:, 6] >> np.uint64(32)).astype(np.int64)
xcc = (real[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
ring = (real[:, 7] >> np.uint64(32)).astype(np.int64)   # points of the needle inside the annulus
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
unit = ((xcc * 8 + se) * 16 + cu) * 4 + simd
q = lambda v, f: np.sort(v)[min(len(v) - 1, int(len(v) * f))]
print('  time line (us after the launch\'s first wave started):')
for (k, label) in ((0, 'wave starts'), (2, 'colour done (core pixel arrived)'), (3, 'in-range mask done (all rows arrived)'), (4, 'labelling done'), (5, 'wave ends')):
    v = r[:, k]
    print('    %-40s min %5.1f median %5.1f p90 %5.1f p99 %5.1f max %5.1f' % (label, v.min(), np.median(v), q(v, 0.9), q(v, 0.99), v.max()))
units = np.unique(unit)
per = np.array([[(unit == u).sum(), r[unit == u, 5].max(), r[unit == u, 0].min()] for u in units])
print('  %d SIMDs hold waves: waves per SIMD min %d median %d max %d | a SIMD\'s last wave ends: min %.1f median %.1f p90 %.1f max %.1f us' % (
    len(units), per[:, 0].min(), np.median(per[:, 0]), per[:, 0].max(), per[:, 1].min(), np.median(per[:, 1]), q(per[:, 1], 0.9), per[:, 1].max()))
for cnt in sorted(set(per[:, 0].astype(int).tolist())):
    sel = per[:, 0] == cnt
    print('    SIMDs with %d waves: %4d, last end median %.1f max %.1f us' % (cnt, sel.sum(), np.median(per[sel, 1]), per[sel, 1].max()))
late = r[:, 5] >= q(r[:, 5], 0.95)
print('  the last 5 %% of the waves to end (%d): start median %.1f us | phases in shader cycles (their median / all waves\' median):' % (late.sum(), np.median(r[late, 0])))
for k in range(5):
    dlt = t[:, k + 1] - t[:, k]
    print('    %-34s %8.0f / %8.0f' % (names[k], np.median(dlt[late]), np.median(dlt)))
print('    dials of the late waves: %s; waves on their SIMDs: %s' % (np.bincount((np.arange(n) % nd)[late], minlength=nd).tolist(),
      np.bincount(np.array([(unit == u).sum() for u in unit[late]])).tolist()))
for at in (5, 10, 15, 20, 25, 30, 35, 40, 45):
    print('    at %2d us: %4d waves not started, %4d running, %4d done' % (at, (r[:, 0] > at).sum(), ((r[:, 0] <= at) & (r[:, 5] > at)).sum(), (r[:, 5] <= at).sum()))
# which waves share a SIMD: wave index inside the workgroup (= dial) and workgroup (= frame) of every SIMD's waves
wvi = np.arange(n) % nd
wgi = np.arange(n) // nd
same = 0
for u in units:
    if len(set(wvi[unit == u].tolist())) == 1:
        same += 1
print('  SIMDs whose waves all have the same index inside their workgroup: %d of %d' % (same, len(units)))
for u in units[:3].tolist() + units[-2:].tolist():
    print('    SIMD %5d: wave indices %s of workgroups %s' % (u, wvi[unit == u].tolist(), wgi[unit == u].tolist()))
cuid = unit // 4
c0 = np.unique(cuid)[:2]
for c in c0:
    print('    CU %4d: workgroups %s' % (c, sorted(set(wgi[cuid == c].tolist()))))
print('  per dial, median shader cycles of each phase and candidates of the exact test:')
for d in range(nd):
    sel = wvi == d
    print('    dial %d: %s | candidates median %d p90 %d max %d | ends median %.1f us' % (d, ' '.join('%6.0f' % np.median(t[sel, k + 1] - t[sel, k]) for k in range(5)),
          np.median(cands[sel]), q(cands[sel], 0.9), cands[sel].max(), np.median(r[sel, 5])))
# inside the last two phases (shader cycles, medians; stamps 3 = closing start, 4 = labelling done, 5 = wave end)
fine = np.zeros((n, 16), np.uint64)
assert _hip.lib().melf_debug_dials_fine(fine.ctypes.data_as(C.c_void_p), n) == 0
fz = fine.astype(np.float64)
ok = (fz[:, :8] > 0).all(axis=1)
print('  inside the last phases (%d waves with every stamp), median shader cycles [dial 1]:' % ok.sum())
seq = [('closing + row masks', t8[:, 3], fz[:, 0]), ('Euler number + labelling of M', fz[:, 0], fz[:, 1]), ('holes: outside flood + relabelling', fz[:, 1], t8[:, 4]),
       ('momentum loop + reductions', fz[:, 2], fz[:, 3]), ('momentum angle (atan)', fz[:, 3], fz[:, 4]), ('ring list', fz[:, 4], fz[:, 5]),
       ('ring angles (atan) + count/min', fz[:, 5], fz[:, 6]), ('trimming keys', fz[:, 6], fz[:, 7]), ('weighted mean + end', fz[:, 7], t8[:, 5])]
for (label, a, b) in seq:
    dlt = (b - a)[ok]
    d1 = (b - a)[ok & (wvi == 1)]
    print('    %-34s %7.0f  [%7.0f]  p90 %7.0f' % (label, np.median(dlt), np.median(d1), q(dlt, 0.9)))
print('  inside the first phases, median shader cycles [p90]: geometry arrived %.0f [%.0f] | pixels requested %.0f [%.0f] | colour core arrived %.0f [%.0f] | colour and bounds %.0f [%.0f]' % (
    np.median(fz[:, 8] - t8[:, 1]), q(fz[:, 8] - t8[:, 1], 0.9), np.median(fz[:, 9] - fz[:, 8]), q(fz[:, 9] - fz[:, 8], 0.9),
    np.median(fz[:, 10] - fz[:, 9]), q(fz[:, 10] - fz[:, 9], 0.9), np.median(t8[:, 2] - fz[:, 10]), q(t8[:, 2] - fz[:, 10], 0.9)))
# the slowest waves of the launch
worst = np.argsort(-r[:, 5])[:6]
print('  their labelling phase (Euler number + labelling of M / holes: outside flood + relabelling): ' + '; '.join('%.0f / %.0f' % (fz[i, 1] - fz[i, 0], t8[i, 4] - fz[i, 1]) for i in worst))
holes = (t8[:, 4] - fz[:, 1]) > 1000
print('  waves that took the hole path: %d of %d (per dial %s); of the 5 %% that end last: %d of %d' % (holes.sum(), n, np.bincount(wvi[holes], minlength=nd).tolist(), (holes & late).sum(), late.sum()))
print('  the six waves that end last: ' + '; '.join('frame %d dial %d ends %.1f us, %d candidates, phases %s' % (wgi[i], wvi[i], r[i, 5], cands[i],
      '/'.join('%.0f' % (t[i, k + 1] - t[i, k]) for k in range(5))) + ', ring points %d' % ring[i] for i in worst))
print('  ring points per wave: median %d p90 %d p99 %d max %d; waves above 256 (the angle cache): %d, above 512: %d' % (np.median(ring), q(ring, 0.9), q(ring, 0.99), ring.max(), (ring > 256).sum(), (ring > 512).sum()))
# hole-path waves: Euler number (x 4), components of M (0 = the Euler number alone proved the hole), isolated pixels
e4 = fine[:, 12].astype(np.int64).astype(np.int32).astype(np.int64)
nc = fine[:, 13].astype(np.int64)
iso = fine[:, 14].astype(np.int64)
from collections import Counter
print('  hole-path waves by (Euler number, components of M labelled first, isolated pixels): %s' % sorted(Counter(zip((e4[holes] // 4).tolist(), nc[holes].tolist(), iso[holes].tolist())).items()))
print('  the other waves by components of M: %s' % sorted(Counter(nc[~holes].tolist()).items()))
