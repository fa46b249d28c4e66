#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build
container only; /root/reference does not exist on the GPU box).

What it commits is DATA only:
  * the reference's own test data files, copied verbatim: the sample JPEGs,
    params.yml and template PNG of sample-images1/2 and the two golden stdout
    files tests/sample-images{1,2}_stdout.txt (the latter of sample-images1 is
    identical to integration-tests/test_all_sample_images.expected_stdout);
  * crop_crc32.json: CRC32 of every decoded meter_rect crop (Pillow /
    libjpeg-turbo), so that JPEG-decoder drift on another box is detected
    instead of silently changing digits;
  * pure_fn_vectors.json: input/output vectors of the reference's pure-Python
    helpers (get_angle_by_vector, determine_value_by_dial_positions,
    HlsColor.get_range), produced by importing the reference with a stub
    `cv2` module (cv2 itself is not installed).
"""
import glob
import json
import os
import random
import shutil
import sys
import types
import zlib

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def copy_data():
    for sd, tpl in (('sample-images1', 'dials_gray.png'), ('sample-images2', 'dials_template.png')):
        dst = os.path.join(HERE, sd)
        os.makedirs(dst, exist_ok=True)
        for f in glob.glob(os.path.join(REF, sd, '*.jpg')) + [os.path.join(REF, sd, 'params.yml'),
                                                             os.path.join(REF, sd, tpl)]:
            shutil.copyfile(f, os.path.join(dst, os.path.basename(f)))
    for f in ('sample-images1_stdout.txt', 'sample-images2_stdout.txt'):
        shutil.copyfile(os.path.join(REF, 'tests', f), os.path.join(HERE, f))


def crop_crcs():
    import yaml
    from PIL import Image
    out = {}
    for sd in ('sample-images1', 'sample-images2'):
        with open(os.path.join(HERE, sd, 'params.yml')) as fp:
            p = yaml.safe_load(fp)
        (x0, y0), (x1, y1) = p['meter_rect']['top_left'], p['meter_rect']['bottom_right']
        for f in sorted(glob.glob(os.path.join(HERE, sd, '*.jpg'))):
            img = np.asarray(Image.open(f).convert('RGB'), dtype=np.uint8)[:, :, ::-1]
            crop = np.ascontiguousarray(img[y0:y1, x0:x1])
            out[sd + '/' + os.path.basename(f)] = zlib.crc32(crop.tobytes())
    with open(os.path.join(HERE, 'crop_crc32.json'), 'w') as fp:
        json.dump(out, fp, indent=0, sort_keys=True)


def pure_fn_vectors():
    sys.modules['cv2'] = types.ModuleType('cv2')  # stub: only pure helpers are called
    sys.path.insert(0, REF)
    from meterelf import _colors, _reading, _utils
    rnd = random.Random(20260101)
    vec = {'angle_by_vector': [], 'value_by_positions': [], 'get_range': []}
    pts = [(0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, 0)]
    for _ in range(300):
        pts.append((rnd.randint(-30, 30) - rnd.choice([0.0, 0.3, 0.4, 0.9]),
                    rnd.randint(-30, 30) - rnd.choice([0.0, 0.4, 0.5, 0.9])))
    for (x, y) in pts:
        vec['angle_by_vector'].append([x, y, _utils.get_angle_by_vector((x, y))])
    names = ['0.0001', '0.001', '0.01', '0.1']
    for k in range(400):
        r = [rnd.uniform(0, 10) for _ in range(4)]
        if k % 3 == 0:  # stress the carry thresholds
            r = [rnd.choice([0.0, 1.99, 2.0, 2.01, 7.99, 8.0, 9.999]) if rnd.random() < .5 else
                 rnd.randint(0, 9) + rnd.choice([0.44, 0.45, 0.46, 0.54, 0.55, 0.56, 0.0, 0.999]) for _ in range(4)]
        v = _reading.determine_value_by_dial_positions(dict(zip(names, r)))
        vec['value_by_positions'].append([r, v])
    for _ in range(200):
        c = [rnd.randint(0, 255) for _ in range(3)]
        g = [rnd.randint(0, 255) for _ in range(3)]
        lo, hi = _colors.HlsColor(*c).get_range(_colors.HlsColor(*g))
        vec['get_range'].append([c, g, [int(v) for v in lo], [int(v) for v in hi]])
    with open(os.path.join(HERE, 'pure_fn_vectors.json'), 'w') as fp:
        json.dump(vec, fp)


if __name__ == '__main__':
    copy_data()
    crop_crcs()
    pure_fn_vectors()
    print('fixtures regenerated in', HERE)
