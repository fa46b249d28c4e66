#!/usr/bin/env python3
"""What do the reference's 304 golden lines actually pin?  (CPU only; the oracle is the only thing that runs.)

OpenCV 3.4.5 is absent, so oracle/melf_oracle.c restates several of its semantics from knowledge of its sources
(SURVEY.md appendix A).  This script flips each of those believed semantics to its plausible alternative
(ORC_OPT_* switches in the oracle) and counts, over all 81 + 223 fixture frames,

  golden_lines_changed   output lines that no longer equal the reference's golden stdout (what the goldens pin)
  records_changed        frames whose record differs AT ALL from the unflipped oracle (status, match, any dial
                         position bit) -- 0 means the fixtures never exercise the semantic

A semantic with golden_lines_changed == 0 is NOT pinned by the reference's goldens: for it the HIP path can only be
claimed equal to the oracle.      python3 tools/oracle_sensitivity.py [--write]   (writes tests/golden/oracle_sensitivity.json)
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as po  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
NOISY = '20180814021310-00-e02.jpg'
VARIANTS = [('hls_variant', 1), ('hls_variant', 2), ('contour_tie', 1), ('mean_form', 1), ('hls_round', 1), ('area_rule', 1),
            ('no_hole_fill', 1), ('erode_border', 1), ('l_integer', 1), ('minmax_last', 1), ('hue_g_first', 1)]
WHAT = {
    ('hls_variant', 1): 'BGR2HLS S formula: scalar-tail form `2 - vmax - vmin` for every pixel',
    ('hls_variant', 2): 'BGR2HLS S formula: SIMD form `2 - (vmax + vmin)` for every pixel (no scalar tail)',
    ('contour_tie', 1): 'largest contour: among equal areas the LAST discovered wins (cv2 list order reversed)',
    ('mean_form', 1): 'cv::mean as sum / N instead of sum * (1. / N) (dial colour core and template mean)',
    ('hls_round', 1): 'cvRound as round-half-away instead of half-to-even in BGR2HLS',
    ('area_rule', 1): 'contourArea = pixel count of the filled contour instead of the polygon through pixel centres',
    ('no_hole_fill', 1): 'drawContours(-1) paints the component only (holes stay open)',
    ('erode_border', 1): 'erode treats pixels outside the image as 0 instead of the neutral default border',
    ('l_integer', 1): 'L = (max + min + 1) >> 1 instead of the float32 path',
    ('minmax_last', 1): 'minMaxLoc returns the last maximum in raster order instead of the first',
    ('hue_g_first', 1): 'hue sector test tries vmax == g before vmax == r (differs on r == g ties)',
}


def load():
    data = {}
    for sd in ('sample-images1', 'sample-images2'):
        params = po.Params(os.path.join(GOLDEN, sd, 'params.yml'))
        with open(os.path.join(GOLDEN, sd + '_stdout.txt')) as fp:
            expected = dict(line.split(': ', 1) for line in fp.read().splitlines())
        crops = []
        for f in sorted(glob.glob(os.path.join(GOLDEN, sd, '*.jpg'))):
            crops.append((os.path.basename(f), np.ascontiguousarray(po.crop_meter(po.decode_bgr(f), params))))
        data[sd] = (params, expected, crops)
    return data


def run(data):
    """-> (lines that differ from the golden stdout, raw records)"""
    bad = 0
    records = []
    for (sd, (params, expected, crops)) in data.items():
        for (name, crop) in crops:
            res = po.process_crop(crop, params)
            line = po.output_line(name, res, params).split(': ', 1)[1]
            exp = expected[name]
            if line != exp:
                ok = False
                if name == NOISY:  # the one declared tolerance (OpenCV's float32-DFT noise in the printed match value)
                    pat = r'UNKNOWN Dials not found \(match val = ([0-9.]+)\)'
                    (g, e) = (re.fullmatch(pat, line), re.fullmatch(pat, exp))
                    ok = bool(g and e and abs(float(g.group(1)) - float(e.group(1))) <= 1e-5 * float(e.group(1)))
                bad += 0 if ok else 1
            records.append((res.status, res.match_x, res.match_y, float(res.match_val), res.failed_dial, res.unreadable_mask,
                            tuple(res.pos[:4]), float(res.value)))
    return bad, records


def audit():
    data = load()
    for name in po.OPTIONS:
        po.set_option(name, 0)
    (base_bad, base) = run(data)
    assert base_bad == 0, 'the unflipped oracle must reproduce all goldens'
    rows = []
    for (name, value) in VARIANTS:
        po.set_option(name, value)
        try:
            (bad, recs) = run(data)
        finally:
            po.set_option(name, 0)
        changed = sum(1 for (a, b) in zip(base, recs) if a != b)
        maxd = max([abs(x - y) for (a, b) in zip(base, recs) if a[0] == 0 and b[0] == 0 for (x, y) in zip(a[6], b[6])] or [0.0])
        rows.append({'switch': '%s=%d' % (name, value), 'what': WHAT[(name, value)], 'golden_lines_changed': bad,
                     'records_changed': changed, 'max_position_shift': round(min(maxd, 10 - maxd), 6)})
    return rows


def main():
    rows = audit()
    print('| believed semantic flipped | golden lines changed (of 304) | records changed at all | largest dial-position shift |')
    print('|---|---|---|---|')
    for r in rows:
        print('| %s (`%s`) | %d | %d | %g |' % (r['what'], r['switch'], r['golden_lines_changed'], r['records_changed'], r['max_position_shift']))
    if '--write' in sys.argv:
        with open(os.path.join(GOLDEN, 'oracle_sensitivity.json'), 'w') as fp:
            json.dump(rows, fp, indent=1)
            fp.write('\n')


if __name__ == '__main__':
    main()
