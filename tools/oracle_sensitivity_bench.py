#!/usr/bin/env python3
"""Do bench.py's TIMED workloads exercise the OpenCV semantics the reference's goldens do not pin?  (round 4's VERDICT, item 7)

tools/oracle_sensitivity.py shows that seven believed semantics of the oracle (S-formula variants, contour tie-break, cv::mean
form, polygon-area rule, erode border, hue sector tie) are noticed by none of the 304 golden lines.  bench.py claims "digits
identical to the oracle" on synthesised frames (configs 3 / 4: fixtures shifted by up to 8 px plus sigma-2 noise; config 5: the
same crops inside 1080p frames, six dials): this script flips each semantic in the oracle and counts, over the first 1024 frames of
config 3, of config 4 and over 64 frames of config 5 -- generated exactly as bench.py generates them, on the GPU's generator --
how many records change.  Zero: the timed workloads do not exercise that belief either.  Non-zero: the bench's parity claim
carries that condition.

    python3 tools/oracle_sensitivity_bench.py gen DIR      (GPU box: writes the workloads' meter crops to DIR)
    python3 tools/oracle_sensitivity_bench.py run DIR OUT  (CPU only: process pool over the host's cores; table to stdout, JSON to OUT)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
UNPINNED = [('hls_variant', 1), ('hls_variant', 2), ('contour_tie', 1), ('mean_form', 1), ('area_rule', 1), ('erode_border', 1), ('hue_g_first', 1)]
PINNED = [('hls_round', 1), ('l_integer', 1), ('no_hole_fill', 1), ('minmax_last', 1)]   # for scale: what the flips the goldens DO notice change here


def gen(out_dir):
    import torch

    import bench
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device('cuda', 0)
    from oracle import pyoracle as po
    for (label, sd, seed) in (('config3', 'sample-images1', 2024), ('config4', 'sample-images2', 2025)):
        base = bench.load_fixture_frames(sd)
        frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), 1024, seed, dev)   # = batch 0 of bench.py's rank 0
        P = po.Params(os.path.join(GOLDEN, sd, 'params.yml'))
        ((x0, y0), (x1, y1)) = P.meter_rect
        np.save(os.path.join(out_dir, label + '.npy'), frames[:, y0:y1, x0:x1].contiguous().cpu().numpy())
        if label == 'config3':
            # config 5's meter crops ARE config 3's (bench.config5_block pastes them at the 1080p meter_rect); its 64-frame gate
            # takes frames spread evenly over the 512-frame batch
            pick = np.unique(np.linspace(0, 511, 64).astype(np.int64))
            np.save(os.path.join(out_dir, 'config5.npy'), frames[torch.from_numpy(pick).to(dev)][:, y0:y1, x0:x1].contiguous().cpu().numpy())
    print('workloads written to', out_dir)


_STATE = {}


def _work(args):
    (label, lo, hi, name, value) = args
    from oracle import pyoracle as po
    for n in po.OPTIONS:
        po.set_option(n, 0)
    if name:
        po.set_option(name, value)
    (params, crops) = _STATE[label]
    out = []
    for i in range(lo, hi):
        r = po.process_crop(crops[i], params)
        nd = params.c_params().ndials
        out.append((r.status, r.match_x, r.match_y, float(r.match_val), r.failed_dial, r.unreadable_mask, tuple(r.pos[:nd]), float(r.value)))
    return out


def run(in_dir, out_file):
    import multiprocessing as mp

    import bench
    from oracle import pyoracle as po
    d5 = bench.config5_params_dir()
    try:
        plist = {'config3': po.Params(os.path.join(GOLDEN, 'sample-images1', 'params.yml')),
                 'config4': po.Params(os.path.join(GOLDEN, 'sample-images2', 'params.yml')),
                 'config5': po.Params(os.path.join(d5, 'params.yml'))}
        for p in plist.values():
            p.load_template()
            p.masks()
        # config 5's oracle params place the meter_rect inside the 1080p frame; the crops are that rect already
        for (label, p) in plist.items():
            _STATE[label] = (p, np.load(os.path.join(in_dir, label + '.npy')))
        ncpu = len(os.sched_getaffinity(0))
        rows = []
        with mp.Pool(ncpu) as pool:
            for label in ('config3', 'config4', 'config5'):
                n = len(_STATE[label][1])
                step = max(1, (n + 4 * ncpu - 1) // (4 * ncpu))
                spans = [(lo, min(n, lo + step)) for lo in range(0, n, step)]

                def records(name, value):
                    return [r for part in pool.map(_work, [(label, lo, hi, name, value) for (lo, hi) in spans]) for r in part]
                base = records(None, 0)
                ok = sum(1 for r in base if r[0] == 0)
                for (name, value) in UNPINNED + PINNED:
                    recs = records(name, value)
                    changed = sum(1 for (a, b) in zip(base, recs) if a != b)
                    digits = sum(1 for (a, b) in zip(base, recs) if (a[0], '%07.3f' % a[7]) != (b[0], '%07.3f' % b[7]))
                    rows.append({'workload': label, 'frames': n, 'frames_read_ok': ok, 'switch': '%s=%d' % (name, value),
                                 'pinned_by_goldens': (name, value) in PINNED, 'records_changed': changed, 'printed_values_changed': digits})
                    print('%-8s %-16s records changed %4d / %d   printed values changed %4d%s' % (
                        label, rows[-1]['switch'], changed, n, digits, '   (a flip the goldens notice: for scale)' if (name, value) in PINNED else ''), flush=True)
    finally:
        import shutil
        shutil.rmtree(d5, ignore_errors=True)
    with open(out_file, 'w') as fp:
        json.dump(rows, fp, indent=1)
        fp.write('\n')


if __name__ == '__main__':
    if len(sys.argv) >= 3 and sys.argv[1] == 'gen':
        gen(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == 'run':
        run(sys.argv[2], sys.argv[3])
    else:
        print(__doc__)
        sys.exit(2)
