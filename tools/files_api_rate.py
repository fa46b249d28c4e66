#!/usr/bin/env python3
"""The file-name C entry points alone (melf_jpeg_process_files_begin / _end, one to three calls in flight), without the Python side
of get_meter_values: path arrays marshalled once, records not converted -- what a compiled host gets.
    python3 tools/files_api_rate.py [chunk]"""
import ctypes as C
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from meterelf_amd import MeterReader, _hip, _params

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = os.path.join(ROOT, 'tests', 'golden', 'sample-images2')
reader = MeterReader(_params.load(os.path.join(d, 'params.yml')))
files = (sorted(glob.glob(os.path.join(d, '*.jpg'))) * 64)[:28 * chunk]
L = _hip.lib()
h = reader.ctx._h
jobs = []
for k in range(0, len(files), chunk):
    part = files[k:k + chunk]
    enc = [os.fsencode(p) for p in part]
    jobs.append(((C.c_char_p * len(part))(*enc), enc, len(part), np.zeros(len(part), _hip.RESULT_DTYPE), np.zeros(len(part), np.int32),
                 C.c_int32(0), C.c_int32(0)))


def begin(j):
    (arr, _enc, n, out, status, H, W) = j
    _hip.check(L.melf_jpeg_process_files_begin(h, arr, n, C.byref(H), C.byref(W), out.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p)))


ROUNDS = int(os.environ.get('FILES_RATE_ROUNDS', '8'))   # passes over the job list per depth
for depth in range(1, _hip.FILES_IN_FLIGHT_MAX + 1):
    stamps = []
    t0 = time.perf_counter()
    for rep in range(ROUNDS):
        inflight = 0
        nxt = 0
        done = 0
        while done < len(jobs):
            while inflight < depth and nxt < len(jobs):
                begin(jobs[nxt]); nxt += 1; inflight += 1
            _hip.check(L.melf_jpeg_process_files_end(h)); inflight -= 1; done += 1
            stamps.append(time.perf_counter())
    dt = time.perf_counter() - t0
    ok = sum(int((j[4] == 0).sum()) for j in jobs)
    gaps = np.diff(np.array(stamps)) * 1e3   # time between consecutive calls' completions: the pipeline's period
    print('%d call(s) in flight, chunks of %d: %d files in %.1f ms = %.0f files/s overall; per call: median %.3f ms (= %.0f files/s), p10 %.3f, p90 %.3f  (%d decoded in the last pass)'
          % (depth, chunk, ROUNDS * len(files), dt * 1e3, ROUNDS * len(files) / dt, np.median(gaps), chunk / np.median(gaps) * 1e3,
             np.percentile(gaps, 10), np.percentile(gaps, 90), ok))
reader.close()
