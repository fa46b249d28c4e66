"""Times melf_jpeg_process_batch (GPU JPEG decode + meter reading) on fixture files: ms per call and per kernel.

    python3 tools/jpeg_timing.py [sample-images2] [n]      (MELF_JPEG_TRACE=1 adds the host/device split)
"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meterelf_amd import MeterReader, _hip, _params
d = os.path.join(ROOT, 'tests', 'golden', sys.argv[1] if len(sys.argv) > 1 else 'sample-images2')
reader = MeterReader(_params.load(os.path.join(d, 'params.yml')))
blobs = [open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(d, '*.jpg')))]
(H, W, ok, _) = _hip.jpeg_probe(blobs[-1])
blobs = [b for b in blobs if _hip.jpeg_probe(b)[:2] == (H, W)]
blobs = (blobs * 20)[:int(sys.argv[2]) if len(sys.argv) > 2 else 1024]
import numpy as np
reader.ctx.jpeg_process_batch(blobs, H, W)  # warm-up: allocates the workspaces
reader.ctx.set_profiling(True); reader.ctx.timings()
t0 = time.perf_counter()
for _ in range(3):
    (recs, status) = reader.ctx.jpeg_process_batch(blobs, H, W)
dt = (time.perf_counter() - t0) / 3
t = reader.ctx.timings()
print('n=%d %.2f ms/call | ' % (len(blobs), dt * 1e3) + '  '.join('%s %.3f' % (k, ms / c) for (k, (ms, c)) in t.items() if c))
assert (status == 0).all()
