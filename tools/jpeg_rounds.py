#!/usr/bin/env python3
"""How many synchronisation rounds the segment-parallel Huffman kernel needs, and how many segment decodes it
repeats (diagnostic build `make -C meterelf_amd/csrc stamp`, which returns them in the status word's high bits)."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_stamp.so'))
import numpy as np

from meterelf_amd import MeterReader, _hip, _params

sd = sys.argv[1] if len(sys.argv) > 1 else 'sample-images2'
d = os.path.join(ROOT, 'tests', 'golden', sd)
reader = MeterReader(_params.load(os.path.join(d, 'params.yml')))
blobs = [open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(d, '*.jpg')))]
(H, W, ok, _) = _hip.jpeg_probe(blobs[-1])
blobs = [b for b in blobs if _hip.jpeg_probe(b)[:2] == (H, W)]
blobs = (blobs * 16)[:1024]  # a full batch, so that the phase times include the contention of a real launch
n = len(blobs)
out = np.zeros((n, H, W, 3), np.uint8)
status = np.zeros(n, np.int32)
(ptrs, sizes, keep) = _hip._file_table(blobs)
L = _hip.lib()
if len(sys.argv) > 2 and sys.argv[2] == 'crop':  # the windowed decode of melf_jpeg_process_batch instead of whole frames
    (_, status) = reader.ctx.jpeg_process_batch(blobs, H, W)
else:
    _hip.check(L.melf_jpeg_decode_batch(reader.ctx._h, ptrs, sizes, n, H, W, out.ctypes.data_as(C.c_void_p), 0, status.ctypes.data_as(C.c_void_p)))
assert (status == 0).all()
st = np.zeros(n, np.uint32)
assert L.melf_debug_jpeg_rounds(st.ctypes.data_as(C.c_void_p), n) == 0
rounds = (st >> 16).astype(int)   # (of a chunked call: the last launch's images, the rest of the array is zero)
redo = (st & 0xffff).astype(int)
per_launch = 256 if (len(sys.argv) > 2 and sys.argv[2] == 'crop' and n > 256) else n   # the pipelined call's chunk size
(rounds, redo) = (rounds[:per_launch], redo[:per_launch])
print('%s: %d files, %d in the last launch | rounds after the first pass: min %d median %d max %d | segments decoded again (%% of segments, summed over rounds): median %d max %d'
      % (sd, n, per_launch, rounds.min(), int(np.median(rounds)), rounds.max(), int(np.median(redo)), redo.max()))
print('  histogram of rounds:', {int(k): int(v) for (k, v) in zip(*np.unique(rounds, return_counts=True))})

n2 = min(n, 8192)
stamps = np.zeros((n2, 8), np.uint64)
assert L.melf_debug_jpeg_stamps(stamps.ctypes.data_as(C.c_void_p), n2) == 0
t = stamps[:, :7].astype(np.float64)
names = ['zero window + tables', 'round 0 (speculative decode)', 'synchronisation rounds', 'prefix scan', 'output pass', 'DC prefix scan + fix-up']
tot = t[:, 6] - t[:, 0]
# a chunked call (melf_jpeg_process_batch: 256 files per launch) leaves the stamps of its LAST launch only
seen = tot > 0
(t, tot) = (t[seen], tot[seen])
seen_idx = np.nonzero(seen)[0]
print('%d workgroups stamped (the call\'s last launch); cycles per workgroup: median %.0f max %.0f' % (len(tot), np.median(tot), tot.max()))
for k in range(6):
    dlt = t[:, k + 1] - t[:, k]
    print('  %-30s median %8.0f (%4.1f %%)  max %8.0f' % (names[k], np.median(dlt), 100 * np.median(dlt) / np.median(tot), dlt.max()))

# the kernel ends with its slowest workgroup: which images are those?
order = np.argsort(-tot)[:6]
print('  slowest workgroups (image of the launch: cycles | per phase | rounds):')
for i in order:
    print('    %4d: %8.0f | %s | %2d rounds' % (i, tot[i], ' '.join('%7.0f' % (t[i, k + 1] - t[i, k]) for k in range(6)), rounds[i] if i < len(rounds) else -1))
if len(tot) > 2:
    print('  correlation of a workgroup\'s cycles with its rounds: %.2f' % np.corrcoef(tot, rounds[:len(tot)])[0, 1])
    print('  cycles: p50 %.0f p90 %.0f p99 %.0f max %.0f' % tuple(np.percentile(tot, [50, 90, 99, 100])))

if hasattr(L, 'melf_debug_jpeg_round_log'):
    log = np.zeros((8, 32, 5), np.uint32)
    if L.melf_debug_jpeg_round_log(log.ctypes.data_as(C.c_void_p)) == 0:
        for im in range(2):
            print('  image %d, per round: segments decoded again / cycles / loop iterations of the slowest lane | of all lanes, two-symbol steps among them' % im)
            for r in log[im]:
                if r[0]:
                    print('    %4d %7d %4d | %6d %6d' % (r[0], r[1], r[2], r[3], r[4]))
