#!/usr/bin/env python3
"""Where get_meter_values spends its time per 1024-file chunk: the library's stages (melf_ctx_files_stats) and the Python side
(meterelf_amd._api.api_stats), with one context and with the fan-out over two / three contexts on this GPU.
    python3 tools/api_host_breakdown.py [devices ...]     default: 0  0,0  0,0,0"""
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meterelf_amd import _api, get_meter_values, release_cached_contexts  # noqa: E402

d = os.path.join(ROOT, 'tests', 'golden', 'sample-images1')
pfile = os.path.join(d, 'params.yml')
files = [f for f in sorted(glob.glob(os.path.join(d, '*.jpg')))][2:]
names = [files[i % len(files)] for i in range(64 * 1024)]
print('os.cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), flush=True)
for devs in (sys.argv[1:] or ['0', '0,0', '0,0,0']):
    os.environ['METERELF_DEVICES'] = devs
    sum(1 for _ in get_meter_values(pfile, names[:4096]))
    for rep in range(2):
        _api.api_stats(reset=True)
        t0 = time.perf_counter()
        n = sum(1 for r in get_meter_values(pfile, names) if r.error is None)
        dt = time.perf_counter() - t0
        st = _api.api_stats(reset=True)
        ch = max(st['chunks'], 1)
        lib = st['library']
        calls = max(sum(x['calls'] for x in lib), 1.0)
        print('METERELF_DEVICES=%-6s %7.0f files/s | per chunk: wall %.2f ms | python: begin %.2f, blocked in end %.2f, convert %.2f | library (%d pipelines, io %d / host %d threads, %d cores / %d devices): read %.2f, turn wait %.2f, enqueue %.2f, gpu wait %.2f'
              % (devs, len(names) / dt, dt / ch * 1e3, st['s_begin'] / ch * 1e3, st['s_end_wait'] / ch * 1e3, st['s_convert'] / ch * 1e3, len(lib),
                 lib[0]['io_threads'] if lib else 0, lib[0]['host_threads'] if lib else 0, lib[0]['cores'] if lib else 0, lib[0]['devices_in_process'] if lib else 0,
                 sum(x['ms_read'] for x in lib) / calls, sum(x['ms_turn_wait'] for x in lib) / calls, sum(x['ms_enqueue'] for x in lib) / calls,
                 sum(x['ms_gpu_wait'] for x in lib) / calls), flush=True)
    release_cached_contexts()
