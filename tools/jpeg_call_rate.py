#!/usr/bin/env python3
"""melf_jpeg_process_batch alone (file bytes in host memory -> records), pointer table built once: what a compiled host
pays per call.  Sweeps the pipeline's chunk plan (MELF_JPEG_CHUNK, read by the library at every call).
    python3 tools/jpeg_call_rate.py [sample dir] [n] [chunk plans, e.g. default 512:all 256:chunk 128,256,320,320]"""
import ctypes as C
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the reading-path / header-parse arrangements (second and third field of a plan) are switches of the DIAGNOSTIC build
os.environ.setdefault('MELF_LIB_PATH', os.path.join(ROOT, 'meterelf_amd', 'csrc', 'libmeterelf_hip_diag.so'))
import numpy as np

from meterelf_amd import MeterReader, _hip, _params

sd = sys.argv[1] if len(sys.argv) > 1 else 'sample-images1'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
plans = sys.argv[3:] or ['default']
d = os.path.join(ROOT, 'tests', 'golden', sd)
reader = MeterReader(_params.load(os.path.join(d, 'params.yml')))
blobs = [open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(d, '*.jpg')))]
(H, W) = _hip.jpeg_probe(blobs[-1])[:2]
blobs = [b for b in blobs if _hip.jpeg_probe(b)[:3] == (H, W, True)]
batch = [blobs[i % len(blobs)] for i in range(n)]
(ptrs, sizes, keep) = _hip._file_table(batch)
out = np.zeros(n, _hip.RESULT_DTYPE)
status = np.zeros(n, np.int32)
L = _hip.lib()
h = reader.ctx._h


def call():
    _hip.check(L.melf_jpeg_process_batch(h, ptrs, sizes, n, H, W, out.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p)))


def select(plan):
    (sizes_, _, read_) = plan.partition(':')   # "128,256:chunk" / "512:all": the reading path per chunk or once per call
    (read_, _, parse_) = read_.partition(':')   # third field: "all" = every header parsed before the first chunk
    for (key, val) in (('MELF_JPEG_CHUNK', '' if sizes_ == 'default' else sizes_), ('MELF_JPEG_READ', read_), ('MELF_JPEG_PARSE', parse_)):
        if val:
            os.environ[key] = val
        else:
            os.environ.pop(key, None)


# the plans take turns, a few calls each, many rounds: a box's drift and its noisy neighbours hit every plan alike
ROUNDS = int(os.environ.get('JPEG_RATE_ROUNDS', '12'))
K = int(os.environ.get('JPEG_RATE_K', '8'))
times = {plan: [] for plan in plans}
ref = None
for rnd in range(ROUNDS):
    for plan in plans:
        select(plan)
        call()
        for _ in range(K):
            t0 = time.perf_counter()
            call()
            times[plan].append(time.perf_counter() - t0)
        assert (status == 0).all()
        if ref is None:
            ref = out.tobytes()
        assert out.tobytes() == ref, plan
for plan in plans:
    t = np.sort(np.array(times[plan])) * 1e3
    print('plan %-34s median %.3f ms  p10 %.3f  min %.3f  per %d-file call = %.0f files/s at the median (%d calls)' % (
        plan, np.median(t), t[len(t) // 10], t[0], n, n / np.median(t) * 1e3, len(t)), flush=True)
reader.close()
