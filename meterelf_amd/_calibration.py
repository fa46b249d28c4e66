"""Dial-centre calibration with the reference's entry points (meterelf/_calibration.py:16-84):

    find_dial_centers(params, files=255) -> List[DialCenter]

The per-frame work (template match of every file, aligned float64 average, HLS, inRange) runs
on the GPU through the C ABI; tracing the four needle-hub contours and fitting ellipses to
their ~40 points each is host work.
"""
import glob
import math
import random
from typing import Iterable, List, Tuple, Union

import numpy as np

from . import _hip
from ._engine import MeterReader, result_to_python
from ._image import imread_bgr
from ._params import Params as _Params
from ._types import DialCenter
from .exceptions import DialsNotFoundError, ImageLoadingError

# where ImageFile.get_bgr_image_t moves the dials' top-left corner (meterelf/_image.py:38-41)
ALIGN_TOP_LEFT = (30, 116)
_BAD_FILENAMES = ['20180814021309-01-e01.jpg', '20180814021310-00-e02.jpg']


def find_dial_centers(params: _Params, files: Union[int, Iterable[str]] = 255) -> List[DialCenter]:
    avg_meter = get_average_meter_image(params, get_files(params, files))
    return find_dial_centers_from_image(params, avg_meter)


def get_files(params: _Params, files: Union[int, Iterable[str]] = 255) -> Iterable[str]:
    if isinstance(files, int):
        return random.sample(get_image_filenames(params), files)
    return files


def get_image_filenames(params: _Params) -> List[str]:
    return [path for path in glob.glob(params.image_glob)
            if all(bad not in path for bad in _BAD_FILENAMES)]


def get_average_meter_image(params: _Params, files: Iterable[str]) -> np.ndarray:
    """Average of the aligned meter crops of `files` (u8, crop shape)."""
    names = list(files)
    if not names:
        raise ValueError("Cannot calculate average of empty sequence")
    frames = []
    for fn in names:
        img = imread_bgr(fn)
        if img is None:
            raise ImageLoadingError(fn)
        frames.append(img)
    if len({f.shape for f in frames}) != 1:
        raise ValueError('calibration frames must share one shape')
    batch = np.stack(frames)
    reader = MeterReader(params)
    try:
        recs = reader.ctx.process_batch(batch)
        for (fn, rec) in zip(names, recs):
            if int(rec['status']) == _hip.FRAME_DIALS_NOT_FOUND:  # get_bgr_image_t -> _find_dials raises
                raise result_to_python(rec, reader.dial_names, fn)[1]
        return reader.ctx.aligned_average(batch, recs['match_x'], recs['match_y'], *ALIGN_TOP_LEFT)
    finally:
        reader.close()


def find_dial_centers_from_image(params: _Params, avg_meter: np.ndarray) -> List[DialCenter]:
    """meterelf/_calibration.py:33-57: the average image is treated as an already cropped frame."""
    reader = MeterReader(params)
    try:
        ctx = reader.ctx
        hls = ctx.bgr2hls(avg_meter)
        (mv, mx, my, _map) = ctx.match_ccoeff(np.ascontiguousarray(hls[:, :, 1])[None])
        if float(mv[0]) < params.dials_match_threshold:
            raise DialsNotFoundError('<average_image>', extra_info={'match val': float(mv[0])})
        (th, tw) = params.dials_template_size
        (x, y) = (int(mx[0]), int(my[0]))
        dials_hls = np.ascontiguousarray(hls[y:y + th, x:x + tw])
        (lo, hi) = params.needle_color.get_range(params.needle_color_range)
        needles_mask = ctx.inrange(dials_hls, list(lo), list(hi))
    finally:
        reader.close()
    dial_centers = []
    for contour in find_external_contours(needles_mask):
        (center, size, _angle) = fit_ellipse(contour)
        (height, width) = size
        diameter = (width + height) / 2.0
        if abs(height - width) / diameter > 0.2:
            raise ValueError('Needle center not circle enough')
        dial_centers.append(DialCenter(center, int(round(diameter))))
    return sorted(dial_centers, key=(lambda c: c.center[0]))


# ---- host geometry: cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) and cv2.fitEllipse ----

_STEPS = [(1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1)]  # E, NE, N, ... (y down)


def _trace_outer_border(img: np.ndarray, sx: int, sy: int) -> List[Tuple[int, int]]:
    """Border following from the raster-first pixel (sx, sy) of a component of the zero-padded
    0/1 image `img`: the square-tracing of Suzuki & Abe as cv2 walks it (neighbours examined
    clockwise starting after the direction we came from); every visit is emitted, so pixels of
    one-pixel-wide parts appear twice -- cv2.fitEllipse sees them twice as well."""
    def nz(x, y):
        return img[y, x] != 0
    # first neighbour: counter-clockwise from west (the pixel left of the start is background)
    d = 4
    first = None
    for _ in range(8):
        d = (d - 1) % 8
        if nz(sx + _STEPS[d][0], sy + _STEPS[d][1]):
            first = d
            break
    if first is None:
        return [(sx, sy)]
    points = []
    (x, y) = (sx, sy)
    came = first  # direction index whose clockwise successors are searched next
    (fx, fy) = (sx + _STEPS[first][0], sy + _STEPS[first][1])
    while True:
        nd = None
        for k in range(1, 9):
            dd = (came + k) % 8
            if nz(x + _STEPS[dd][0], y + _STEPS[dd][1]):
                nd = dd
                break
        points.append((x, y))
        (nx, ny) = (x + _STEPS[nd][0], y + _STEPS[nd][1])
        if (nx, ny) == (sx, sy) and (x, y) == (fx, fy):
            break
        (x, y) = (nx, ny)
        came = (nd + 4) % 8
    return points


def find_external_contours(mask: np.ndarray) -> List[np.ndarray]:
    """Outer borders of the 8-connected components of `mask` that do not lie inside another
    component's hole, as (n, 2) int arrays of (x, y); list order as cv2 returns it (last found first)."""
    (h, w) = mask.shape
    img = np.zeros((h + 2, w + 2), np.int8)
    img[1:-1, 1:-1] = (mask != 0)
    # pixels 4-connected to the outside through the background: anything else that is background
    # is a hole, and a component is external iff its raster-first pixel's left neighbour is outside
    outside = np.zeros_like(img, dtype=bool)
    stack = [(0, 0)]
    outside[0, 0] = True
    while stack:
        (x, y) = stack.pop()
        for (dx, dy) in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            (nx, ny) = (x + dx, y + dy)
            if 0 <= nx < w + 2 and 0 <= ny < h + 2 and not outside[ny, nx] and img[ny, nx] == 0:
                outside[ny, nx] = True
                stack.append((nx, ny))
    seen = np.zeros_like(img, dtype=bool)
    contours = []
    for y in range(1, h + 1):
        for x in range(1, w + 1):
            if img[y, x] and not seen[y, x]:
                # label the whole component
                comp = [(x, y)]
                seen[y, x] = True
                i = 0
                while i < len(comp):
                    (cx, cy) = comp[i]
                    i += 1
                    for (dx, dy) in _STEPS:
                        (nx, ny) = (cx + dx, cy + dy)
                        if img[ny, nx] and not seen[ny, nx]:
                            seen[ny, nx] = True
                            comp.append((nx, ny))
                if outside[y, x - 1]:
                    pts = _trace_outer_border(img, x, y)
                    contours.append(np.array([(px - 1, py - 1) for (px, py) in pts], dtype=np.int32))
    return contours[::-1]


def fit_ellipse(points: np.ndarray) -> Tuple[Tuple[float, float], Tuple[float, float], float]:
    """cv2.fitEllipse (OpenCV 3.4): conic A..E by least squares around the centroid, centre from
    the conic's gradient, second least-squares fit of A..C about that centre, then axes.
    Returns (center, (width, height), angle) with width <= height, float32-rounded like cv2's
    RotatedRect."""
    pts = np.asarray(points, dtype=np.float32)
    n = len(pts)
    if n < 5:
        raise ValueError('There should be at least 5 points to fit the ellipse')
    centroid = (pts.sum(axis=0, dtype=np.float32) / np.float32(n)).astype(np.float32)
    rel = (pts - centroid).astype(np.float64)
    (px, py) = (rel[:, 0], rel[:, 1])
    (conic, *_rest) = np.linalg.lstsq(np.column_stack([-px * px, -py * py, -px * py, px, py]),
                                      np.full(n, 10000.0), rcond=None)
    (off, *_rest) = np.linalg.lstsq(np.array([[2 * conic[0], conic[2]], [conic[2], 2 * conic[1]]]),
                                    np.array([conic[3], conic[4]]), rcond=None)
    (qx, qy) = (px - off[0], py - off[1])
    (abc, *_rest) = np.linalg.lstsq(np.column_stack([qx * qx, qy * qy, qx * qy]), np.ones(n), rcond=None)
    theta = -0.5 * math.atan2(abc[2], abc[1] - abc[0])
    t = abc[2] / math.sin(-2.0 * theta) if abs(abc[2]) > 1e-8 else abc[1] - abc[0]
    axes = []
    for v in (abs(abc[0] + abc[1] - t), abs(abc[0] + abc[1] + t)):
        axes.append(math.sqrt(2.0 / v) if v > 1e-8 else v)
    center = (float(np.float32(off[0]) + centroid[0]), float(np.float32(off[1]) + centroid[1]))
    (width, height) = (float(np.float32(axes[0] * 2)), float(np.float32(axes[1] * 2)))
    angle = float(np.float32(90 + theta * 180 / math.pi))
    if width > height:
        (width, height) = (height, width)
        angle = float(np.float32(theta * 180 / math.pi))
    return (center, (width, height), angle)
