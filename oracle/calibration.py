"""CPU restatement of the reference's calibration routine (meterelf/_calibration.py:16-84,
meterelf/_image.py:34-44, meterelf/_utils.py:64-88).  TEST INFRASTRUCTURE ONLY, like the rest
of oracle/.  Pinned by the reference's own golden, tests/test_meterelf.py:118-144
(EXPECTED_CENTER_DATA: four dial centres within 0.05 px, exact diameters).
"""
import ctypes as C
import math

import numpy as np

from . import pyoracle as po

# hard-coded alignment target of ImageFile.get_bgr_image_t (meterelf/_image.py:38-41)
ALIGN_X, ALIGN_Y = 30, 116


def translate(img, dx, dy):
    """cv2.warpAffine(img, [[1,0,dx],[0,1,dy]], (w,h)) for integer dx, dy: exact shift, zero border."""
    (h, w) = img.shape[:2]
    out = np.zeros_like(img)
    (xs0, xs1) = (max(0, -dx), min(w, w - dx))
    (ys0, ys1) = (max(0, -dy), min(h, h - dy))
    if xs1 > xs0 and ys1 > ys0:
        out[ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx] = img[ys0:ys1, xs0:xs1]
    return out


def find_dials(crop_bgr, params):
    """ImageFile._find_dials on a meter_rect crop -> (hls, x, y, max_val); None if below threshold."""
    hls = po.bgr2hls(crop_bgr, params.hue_shift)
    (mv, x, y, _m) = po.match_ccoeff(hls[:, :, 1], params.load_template())
    return (hls, x, y, mv)


def aligned_crop(crop_bgr, params):
    """get_bgr_image_t (meterelf/_image.py:34-44)."""
    (_hls, x, y, mv) = find_dials(crop_bgr, params)
    if mv < params.match_threshold:
        raise ValueError('Dials not found (match val = {})'.format(mv))
    return translate(crop_bgr, ALIGN_X - x, ALIGN_Y - y)


def average_image(crops):
    """calculate_average_of_norm_images + denormalize_image (meterelf/_utils.py:64-88): the
    float64 operations in the reference's order."""
    it = iter(crops)
    p = next(it).astype(np.float64) / 255.0
    n = 2
    for img in it:
        p = p * ((n - 1) / n) + ((img.astype(np.float64) / 255.0) / n)
        n += 1
    return ((p * 255.0) + 0.5).astype(np.uint8)


def external_contours(binimg):
    """cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) -> list of (n,2) int arrays, cv2's order
    (reverse of discovery)."""
    binimg = np.ascontiguousarray(binimg, dtype=np.uint8)
    (h, w) = binimg.shape
    pts = np.zeros((h * w * 4 + 16, 2), np.int32)
    counts = np.zeros(h * w // 2 + 4, np.int32)
    L = po.lib()
    L.orc_external_contours.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    n = L.orc_external_contours(po._ptr(binimg), h, w, po._ptr(pts), len(pts), po._ptr(counts), len(counts))
    out = []
    o = 0
    for k in range(n):
        out.append(pts[o:o + counts[k]].copy())
        o += counts[k]
    return out[::-1]


def fit_ellipse(points):
    """cv2.fitEllipse of OpenCV 3.4 (imgproc/src/shapedescr.cpp, the general-conic fit in three
    least-squares stages) -> ((cx, cy), (width, height), angle).  Needs >= 5 points."""
    pts = np.asarray(points, dtype=np.float32)
    n = len(pts)
    if n < 5:
        raise ValueError('fitEllipse needs at least 5 points')
    c = np.float32(pts.sum(axis=0, dtype=np.float32) / np.float32(n))
    p = (pts - c).astype(np.float32)
    (px, py) = (p[:, 0].astype(np.float64), p[:, 1].astype(np.float64))
    A = np.stack([-px * px, -py * py, -px * py, px, py], axis=1)
    b = np.full(n, 10000.0)
    gfp = np.linalg.lstsq(A, b, rcond=None)[0]
    A2 = np.array([[2 * gfp[0], gfp[2]], [gfp[2], 2 * gfp[1]]])
    rp = np.zeros(5)
    rp[:2] = np.linalg.lstsq(A2, np.array([gfp[3], gfp[4]]), rcond=None)[0]
    A3 = np.stack([(px - rp[0]) ** 2, (py - rp[1]) ** 2, (px - rp[0]) * (py - rp[1])], axis=1)
    g3 = np.linalg.lstsq(A3, np.ones(n), rcond=None)[0]
    min_eps = 1e-8
    rp[4] = -0.5 * math.atan2(g3[2], g3[1] - g3[0])
    if abs(g3[2]) > min_eps:
        t = g3[2] / math.sin(-2.0 * rp[4])
    else:
        t = g3[1] - g3[0]
    rp[2] = abs(g3[0] + g3[1] - t)
    if rp[2] > min_eps:
        rp[2] = math.sqrt(2.0 / rp[2])
    rp[3] = abs(g3[0] + g3[1] + t)
    if rp[3] > min_eps:
        rp[3] = math.sqrt(2.0 / rp[3])
    center = (float(np.float32(rp[0]) + c[0]), float(np.float32(rp[1]) + c[1]))
    (width, height) = (float(np.float32(rp[2] * 2)), float(np.float32(rp[3] * 2)))
    angle = float(np.float32(90 + rp[4] * 180 / math.pi))
    if width > height:
        (width, height) = (height, width)
        angle = float(np.float32(rp[4] * 180 / math.pi))  # cv2 leaves box.angle unset (0) + ... in this branch
    return (center, (width, height), angle)


def find_dial_centers_from_image(avg_crop, params):
    """meterelf/_calibration.py:33-57 -> sorted list of ((cx, cy), diameter)."""
    (hls, x, y, mv) = find_dials(avg_crop, params)
    if mv < params.match_threshold:
        raise ValueError('Dials not found (match val = {})'.format(mv))
    (th, tw) = params.template_size
    dials_hls = np.ascontiguousarray(hls[y:y + th, x:x + tw])
    nc, nr = params.needle_color, params.needle_color_range
    lo = np.array([max(c - r, 0) for (c, r) in zip(nc, nr)], np.int32)
    hi = np.array([min(c + r, 255) for (c, r) in zip(nc, nr)], np.int32)
    mask = np.zeros((th, tw), np.uint8)
    po.lib().orc_inrange(po._ptr(dials_hls), th, tw, po._ptr(lo), po._ptr(hi), po._ptr(mask))
    centers = []
    for contour in external_contours(mask):
        (center, (a, b), _angle) = fit_ellipse(contour)
        (height, width) = (a, b)  # the reference unpacks size as (height, width); only their sum / difference matter
        diameter = (width + height) / 2.0
        if abs(height - width) / diameter > 0.2:
            raise ValueError('Needle center not circle enough')
        centers.append((center, int(round(diameter))))
    return sorted(centers, key=lambda c: c[0][0])


def find_dial_centers(params, files):
    crops = (aligned_crop(po.crop_meter(po.decode_bgr(f), params), params) for f in files)
    return find_dial_centers_from_image(average_image(crops), params)
