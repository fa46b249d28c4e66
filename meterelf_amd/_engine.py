"""MeterReader: the batched GPU engine behind the reference-shaped API.

One MeterReader owns one melf_ctx (one GPU).  Frames go in as uint8 BGR arrays
(N, H, W, 3); records come back as a numpy structured array (_hip.RESULT_DTYPE)
and are turned into the reference's dicts / exceptions by result_to_python().
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _debug, _hip
from ._params import Params
from ._types import Rect
from .exceptions import (
    DialAngleDeterminingError, DialsNotFoundError, ImageProcessingError, NeedleContoursNotFoundError)


def load_template(params: Params) -> np.ndarray:
    """cv2.imread(dials_file, IMREAD_GRAYSCALE) (reference: meterelf/_image.py:72-81).

    A grey file is taken as it is.  For a colour file OpenCV 3.4 lets the file format's own library make the grey image:
    libpng for PNG (png_set_rgb_to_gray with 0.299 / 0.587: libpng 1.6 truncates them to the 15-bit coefficients 9797 / 19234 and
    gives blue the rest, 3737, and does NOT round the sum -- checked against libpng itself, tests/test_host_logic.py; palettes
    expanded to RGB first), libjpeg for JPEG (the decoder is asked for JCS_GRAYSCALE: the Y plane itself); every other format is
    decoded to BGR and goes through cvtColor's 14-bit weights (R 4899, G 9617, B 1868).  Pillow's convert('L') is none of
    these (other weights, truncation)."""
    from PIL import Image
    try:
        with Image.open(params.dials_file) as im:
            fmt = (im.format or '').upper()
            if im.mode == 'L':
                template = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
            elif im.mode == '1':
                template = np.ascontiguousarray(np.asarray(im.convert('L'), dtype=np.uint8))
            elif fmt == 'JPEG' and im.mode in ('RGB', 'YCbCr'):
                im.draft('L', im.size)      # libjpeg's grayscale output: the luma plane, no colour conversion
                template = np.ascontiguousarray(np.asarray(im.convert('L'), dtype=np.uint8))
            else:
                rgb = np.asarray(im.convert('RGB'), dtype=np.int64)
                (r, g, b) = (rgb[:, :, 0], rgb[:, :, 1], rgb[:, :, 2])
                if fmt == 'PNG':
                    grey = (r * 9797 + g * 19234 + b * 3737) >> 15
                else:
                    grey = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14
                template = np.ascontiguousarray(grey.astype(np.uint8))
    except Exception:
        raise IOError("Cannot read dials template: {}".format(params.dials_file))
    assert template.shape == params.dials_template_size
    return template


def make_blob(params: Params, meter_rect: Optional[Rect] = None) -> np.ndarray:
    return _hip.pack_blob(params.to_c(meter_rect), load_template(params))


def result_to_python(rec, dial_names: List[str], filename: str = '') -> Tuple[Dict[str, float], Optional[ImageProcessingError]]:
    """One record -> (meter_values dict, error) exactly as get_meter_value would
    return / raise (reference: meterelf/_reading.py:98-115, _image.py:62-64)."""
    status = int(rec['status'])
    if status == _hip.FRAME_OK:
        values = {name: float(rec['pos'][i]) for (i, name) in enumerate(dial_names)}
        if len(dial_names) == 4:
            values['value'] = float(rec['value'])
        return values, None
    if status == _hip.FRAME_DIALS_NOT_FOUND:
        return {}, DialsNotFoundError(filename, extra_info={'match val': float(rec['match_val'])})
    if status == _hip.FRAME_NEEDLE_CONTOURS_NOT_FOUND:
        return {}, NeedleContoursNotFoundError(extra_info={'dial': dial_names[int(rec['failed_dial'])]})
    if status == _hip.FRAME_ANGLE_UNDETERMINED:
        mask = int(rec['unreadable_mask'])
        bad = [name for (i, name) in enumerate(dial_names) if mask >> i & 1]
        extra_info = {}
        if _debug.DEBUG:
            # meterelf/_reading.py:98-104: the dials that WERE read, sorted by name, before 'unreadable dials'
            read = sorted((name, float(rec['pos'][i])) for (i, name) in enumerate(dial_names) if not mask >> i & 1)
            extra_info['dial positions'] = ' (' + ' | '.join('{}: {:.2f}'.format(k, v) for (k, v) in read) + ')'
        extra_info['unreadable dials'] = ', '.join(bad)
        return {}, DialAngleDeterminingError(filename, extra_info=extra_info)
    raise RuntimeError('unknown frame status {}'.format(status))


def records_to_python(records: np.ndarray, ok, dial_names: List[str], filenames: List[str]) -> list:
    """result_to_python over a whole record array: one (meter_values, error) per record, None where `ok[i]` is
    false.  The columns are converted to Python objects once per array instead of field by field per record
    (numpy scalar access costs more than everything else the API does per file)."""
    status = records['status'].tolist()
    pos = records['pos'][:, :len(dial_names)].tolist()
    value = records['value'].tolist()
    four = len(dial_names) == 4
    out: list = []
    for i in range(len(status)):
        if not ok[i]:
            out.append(None)
        elif status[i] == _hip.FRAME_OK:
            values = dict(zip(dial_names, pos[i]))
            if four:
                values['value'] = value[i]
            out.append((values, None))
        else:
            out.append(result_to_python(records[i], dial_names, filenames[i]))
    return out


def records_to_items(records: np.ndarray, ok, dial_names: List[str], filenames: List[str], make) -> list:
    """records_to_python fused with the construction of the API's result objects: make(filename, value, error,
    meter_values) per record, None where `ok[i]` is false.  The common case -- four dials, the frame read -- is ONE
    list comprehension over the columns (the per-file cost of get_meter_values is what bounds it on long lists)."""
    status = records['status'].tolist()
    value = records['value'].tolist()
    if len(dial_names) != 4:
        return [None if not o else make(f, mv.get('value'), err, mv)
                for (f, o, (mv, err)) in zip(filenames, ok, ((result_to_python(records[i], dial_names, filenames[i]) if ok[i] else ({}, None))
                                                             for i in range(len(status))))]
    (n0, n1, n2, n3) = dial_names
    (p0, p1, p2, p3) = (records['pos'][:, k].tolist() for k in range(4))
    good = _hip.FRAME_OK
    out = [make(f, v, None, {n0: a, n1: b, n2: c, n3: d, 'value': v}) if (o and s == good) else None
           for (f, o, s, v, a, b, c, d) in zip(filenames, ok, status, value, p0, p1, p2, p3)]
    for (i, item) in enumerate(out):
        if item is None and ok[i]:
            (mv, err) = result_to_python(records[i], dial_names, filenames[i])
            out[i] = make(filenames[i], mv.get('value'), err, mv)
    return out


class MeterReader:
    def __init__(self, params: Params, device: int = 0, blob: Optional[np.ndarray] = None, ctx: Optional['_hip.Context'] = None) -> None:
        self.params = params
        self.device = device
        self.dial_names = params.dial_names
        self.blob = blob if blob is not None else make_blob(params)
        self.ctx = ctx if ctx is not None else _hip.Context(self.blob, device)   # ctx: created elsewhere from the same blob (broadcast)
        self._crop_ctx: Dict[Tuple[int, int], _hip.Context] = {}
        self._begun: list = []      # read_jpeg_paths_begin: path lists in flight, oldest first
        self._collected: list = []  # their results taken out of the library early (drain_jpeg_paths), oldest first

    def close(self) -> None:
        self.ctx.close()
        for c in self._crop_ctx.values():
            c.close()
        self._crop_ctx.clear()

    def read_frames(self, frames: np.ndarray) -> np.ndarray:
        """Full camera frames (N, H, W, 3) BGR -> records."""
        return self.ctx.process_batch(frames)

    def read_crops(self, crops: np.ndarray) -> np.ndarray:
        """Already meter_rect-cropped images (the reference's bgr_image injection)."""
        (_n, h, w, _c) = crops.shape
        ctx = self._crop_ctx.get((h, w))
        if ctx is None:
            blob = make_blob(self.params, Rect((0, 0), (w, h)))
            ctx = self._crop_ctx[(h, w)] = _hip.Context(blob, self.device)
        return ctx.process_batch(crops)

    def read_jpeg_files(self, files: List[bytes]) -> List[Optional[np.void]]:
        """JPEG files' bytes -> records, decoded and read on the GPU (melf_jpeg_process_batch).
        None for a file the GPU decoder does not take (not a baseline JPEG, corrupt): the caller
        decodes that one on the host."""
        out: List[Optional[np.void]] = [None] * len(files)
        groups: Dict[Tuple[int, int], List[int]] = {}
        (hs, ws, oks) = _hip.jpeg_probe_batch(files)
        for i in np.flatnonzero(oks):
            groups.setdefault((int(hs[i]), int(ws[i])), []).append(int(i))
        for ((h, w), idxs) in groups.items():
            (recs, status) = self.ctx.jpeg_process_batch([files[i] for i in idxs], h, w)
            for (k, i) in enumerate(idxs):
                if status[k] == _hip.JPEG_OK:
                    out[i] = recs[k]
        return out

    def read_jpeg_paths_batch(self, paths: List[str]) -> Tuple[np.ndarray, np.ndarray]:
        """File names -> (records, ok) (melf_jpeg_process_files: the library reads the files itself).  ok[i] is
        false for a file the GPU decoder does not take (unreadable, not a baseline JPEG, corrupt): the caller's
        host branch."""
        records = np.zeros(len(paths), _hip.RESULT_DTYPE)
        ok = np.zeros(len(paths), bool)
        todo = np.arange(len(paths))
        while len(todo):
            (recs, status, _hw) = self.ctx.jpeg_process_files([paths[i] for i in todo])
            status = np.asarray(status)
            good = status == _hip.JPEG_OK
            records[todo[good]] = recs[good]
            ok[todo[good]] = True
            again = todo[status == _hip.JPEG_SIZE_MISMATCH]
            if len(again) == len(todo):
                break  # cannot happen (the first accepted file defines the size); never loop forever
            todo = again
        return records, ok

    def read_jpeg_paths_begin(self, paths: List[str]) -> None:
        """read_jpeg_paths_batch in two halves: the library works on `paths` on its own thread until
        read_jpeg_paths_end() collects (records, ok) of the OLDEST list begun.  Up to _hip.FILES_IN_FLIGHT_MAX lists may be in
        flight (a later one's files are read while an earlier one decodes); nothing else may use this reader while any is --
        drain_jpeg_paths() first."""
        paths = list(paths)
        self.ctx.jpeg_process_files_begin(paths)
        self._begun.append(paths)

    def jpeg_paths_in_flight(self) -> int:
        return len(self._begun) + len(self._collected)

    def _collect_one(self):
        paths = self._begun.pop(0)
        (recs, status, _hw) = self.ctx.jpeg_process_files_end()
        status = np.asarray(status)
        ok = status == _hip.JPEG_OK
        records = np.zeros(len(paths), _hip.RESULT_DTYPE)
        records[ok] = recs[ok]
        return (paths, records, ok, np.flatnonzero(status == _hip.JPEG_SIZE_MISMATCH))

    def drain_jpeg_paths(self) -> None:
        """Waits for every list in flight and keeps the results for read_jpeg_paths_end(): afterwards the context is
        free for other calls (host-decoded frames, files of another frame size)."""
        while self._begun:
            self._collected.append(self._collect_one())

    def discard_jpeg_paths(self) -> None:
        """Waits for every list in flight and forgets the results (a consumer that stopped early)."""
        self.drain_jpeg_paths()
        self._collected.clear()

    def read_jpeg_paths_end(self) -> Tuple[np.ndarray, np.ndarray]:
        (paths, records, ok, again) = self._collected.pop(0) if self._collected else self._collect_one()
        if len(again):  # files of another frame size: their own call(s), now -- with the context to ourselves
            self.drain_jpeg_paths()
            (r2, ok2) = self.read_jpeg_paths_batch([paths[i] for i in again])
            records[again] = r2
            ok[again] = ok2
        return records, ok

    def read_jpeg_paths(self, paths: List[str]) -> List[Optional[np.void]]:
        """The same as a list: a record, or None where the caller has to decode on the host."""
        (records, ok) = self.read_jpeg_paths_batch(paths)
        return [records[i] if ok[i] else None for i in range(len(paths))]

    def read_many(self, images: List[np.ndarray], cropped: Optional[List[bool]] = None) -> List[np.void]:
        """Heterogeneous list of frames: grouped by shape, one batched call per group."""
        cropped = cropped or [False] * len(images)
        groups: Dict[Tuple[bool, Tuple[int, ...]], List[int]] = {}
        for (i, img) in enumerate(images):
            groups.setdefault((cropped[i], img.shape), []).append(i)
        out: List[Optional[np.void]] = [None] * len(images)
        for ((is_crop, _shape), idxs) in groups.items():
            batch = np.stack([images[i] for i in idxs])
            recs = self.read_crops(batch) if is_crop else self.read_frames(batch)
            for (k, i) in enumerate(idxs):
                out[i] = recs[k]
        return out  # type: ignore


_readers: Dict[int, MeterReader] = {}


def get_reader(params: Params) -> MeterReader:
    """Per-Params cache, like the reference's id(params)-keyed caches
    (meterelf/_image.py:69-81, meterelf/_dial_data.py:11-19)."""
    r = _readers.get(id(params))
    if r is None or r.params is not params:
        r = _readers[id(params)] = MeterReader(params)
    return r
