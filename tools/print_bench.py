#!/usr/bin/env python3
"""Short human summary of a bench.py JSON line:   python3 tools/print_bench.py gpurun_out/bench.json"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith('{')][-1])
r = d.get('roofline') or {}
print('value %.0f frames/s  ms/step %.4f  n_gpus %s  rccl_ranks %s' % (d['value'], d['ms_per_step'], d['n_gpus'], d.get('rccl_ranks')))
print('  kernels ms', d.get('kernel_ms'), '| k_match %.4f ms frac %.4f' % (r.get('avg_launch_ms', 0), r.get('frac', 0)))
s = d.get('sustained')
if s:
    print('  sustained %.4f ms/step over %.2f s (one lane %s), k_match %.4f ms frac %.4f' % (s['ms_per_step'], s['seconds'], s.get('single_lane_ms_per_step'), s['k_match_avg_launch_ms'], s['k_match_frac']))
t = d.get('two_streams')
if t:
    print('  two caller streams %.4f ms/step  %.0f frames/s  identical %s' % (t['ms_per_step'], t['frames_per_s'], t.get('records_identical_to_timed_region', t.get('records_identical_to_single_stream'))))
t = d.get('single_lane')
if t:
    print('  one lane (no hint) %.4f ms/step  %.0f frames/s  identical %s' % (t['ms_per_step'], t['frames_per_s'], t['records_identical_to_timed_region']))
t = d.get('resident_hint')
if t:
    print('  one stream + resident hint %.4f ms/step  %.0f frames/s  identical %s' % (t['ms_per_step'], t['frames_per_s'], t.get('records_identical_to_timed_region', t.get('records_identical_to_headline'))))
f = d.get('fused_mask')
if f:
    print('  fused (config 2) %.4f ms  %.0f GB/s  frac %.4f  | two streams %s' % (f['roofline']['avg_launch_ms'], f['roofline']['achieved'], f['roofline']['frac'], f.get('two_streams')))
c = d.get('config4')
if c:
    print('  config4 %.0f frames/s  %.4f ms/step  kernels %s  k_match frac %.4f  mism %s' % (
        c['frames_per_s'], c['ms_per_step'], c['kernel_ms'], c['roofline']['frac'], (c.get('cpu_baseline') or {}).get('parity_mismatches_vs_gpu')))
    if c.get('two_streams'):
        print('    two streams %.4f ms/step  %.0f frames/s' % (c['two_streams']['ms_per_step'], c['two_streams']['frames_per_s']))
    if c.get('single_lane'):
        print('    one lane %.4f ms/step  %.0f frames/s' % (c['single_lane']['ms_per_step'], c['single_lane']['frames_per_s']))
c = d.get('config5')
if c:
    fm = c['fused_mask']['roofline']
    fp = c['full_path']
    print('  config5 fused %.4f ms %.0f GB/s frac %.4f | full %.0f frames/s %.4f ms/step kernels %s' % (
        fm['avg_launch_ms'], fm['achieved'], fm['frac'], fp['frames_per_s'], fp['ms_per_step'], fp['kernel_ms']))
h = d.get('host_fed')
if h:
    print('  host_fed %.0f frames/s  %.2f ms/call  crop PCIe %.1f GB/s' % (h['frames_per_s'], h['ms_per_call'], h['pcie_GBps_crop_bytes']))
j = d.get('jpeg_decode')
if j:
    print('  jpeg %.0f files/s  %.2f ms/call  kernels per call %s' % (j['files_per_s'], j['ms_per_call'], j.get('kernel_ms_per_call') or j.get('kernel_ms')))
    if j.get('get_meter_values'):
        print('    get_meter_values(file names): %.0f files/s over %d files' % (j['get_meter_values']['files_per_s'], j['get_meter_values']['files']))
c = d.get('cpu_baseline')
if c:
    print('  cpu %.1f frames/s (1 core), %.0f (%d cores), parity mismatches %d' % (c['value'], c['all_cores']['value'], c['all_cores']['cores'], c['parity_mismatches_vs_gpu']))
