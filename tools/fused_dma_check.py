#!/usr/bin/env python3
"""MELF_FUSED_CONFIG=6 (pixel rows through LDS-DMA) against the default fused-mask launch: masks compared byte for byte over
several frame shapes and batch sizes, then time per launch (buffers rotating beyond the Infinity Cache).
    python3 tools/fused_dma_check.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import hashlib
    import numpy as np
    from meterelf_amd import _engine, _hip, _params
    ctx = _hip.Context(_engine.make_blob(_params.load(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml'))), 0)
    rng = np.random.default_rng(7)
    for (n, H, W) in ((3, 640, 480), (2, 480, 640), (5, 1080, 1920), (2, 37, 64), (4, 100, 160), (1, 1, 16), (7, 33, 1024), (2, 700, 48), (9, 64, 4096)):
        frames = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
        frames[:, H // 3:H // 3 + 9, :, :] = (rng.integers(0, 40), 90, 200)     # a band that is in range for some hues
        m = ctx.hls_inrange_close(frames)
        print('%d x %d x %d: %s ones %d' % (n, H, W, hashlib.sha1(m.tobytes()).hexdigest()[:16], int((m > 0).sum())))
    sys.exit(0)
outs = {}
for cfg in ('', '6'):
    env = dict(os.environ)
    env.pop('MELF_FUSED_CONFIG', None)
    if cfg:
        env['MELF_FUSED_CONFIG'] = cfg
    outs[cfg] = subprocess.run([sys.executable, __file__, 'child'], env=env, stdout=subprocess.PIPE, text=True).stdout
    print('config %s:\n%s' % (cfg or 'default', outs[cfg]))
print('MASKS IDENTICAL' if outs[''] == outs['6'] and outs[''].count('\n') == 9 else 'MASKS DIFFER')
for (hw, b, nb) in (('640x480', 256, 4), ('1080x1920', 512, 1)):
    for cfg in ('', '6', '', '6'):
        env = dict(os.environ)
        env.pop('MELF_FUSED_CONFIG', None)
        if cfg:
            env['MELF_FUSED_CONFIG'] = cfg
        o = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_stage.py'), 'fused', '--iters', '60', '--hw', hw, '--batch', str(b), '--nbuf', str(nb)],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        print('%s B=%d config %-7s %s' % (hw, b, cfg or 'default', o.strip()))
