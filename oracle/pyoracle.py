"""ctypes driver of the CPU oracle (oracle/melf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by meterelf_amd/.

It restates the reference's host-side flow independently of the product:
params.yml parsing (meterelf/_params.py:30-81), JPEG decode + meter_rect
crop (meterelf/_image.py:46-55), error messages (meterelf/exceptions.py:21-52)
and the CLI output line (meterelf/_main.py:16-22).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle.so')
MAX_DIALS = 8

OK, DIALS_NOT_FOUND, NEEDLE_CONTOURS_NOT_FOUND, ANGLE_UNDETERMINED = range(4)


class OrcDial(C.Structure):
    _fields_ = [('cx', C.c_double), ('cy', C.c_double),
                ('range_h', C.c_int32), ('range_l', C.c_int32), ('range_s', C.c_int32),
                ('negative_momentum', C.c_int32), ('angle_of_zero', C.c_double)]


class OrcParams(C.Structure):
    _fields_ = [('th', C.c_int32), ('tw', C.c_int32), ('hue_shift', C.c_int32),
                ('ndials', C.c_int32), ('match_threshold', C.c_double),
                ('dial', OrcDial * MAX_DIALS)]


class OrcResult(C.Structure):
    _fields_ = [('status', C.c_int32), ('match_x', C.c_int32), ('match_y', C.c_int32),
                ('failed_dial', C.c_int32), ('unreadable_mask', C.c_uint32),
                ('match_val', C.c_float),
                ('pos', C.c_double * MAX_DIALS), ('angle', C.c_double * MAX_DIALS),
                ('value', C.c_double),
                ('dial_color', (C.c_int32 * 3) * MAX_DIALS),
                ('n_needle', C.c_int32 * MAX_DIALS), ('n_outer', C.c_int32 * MAX_DIALS),
                ('n_kept', C.c_int32 * MAX_DIALS), ('contour_area', C.c_double * MAX_DIALS)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, 'melf_oracle.c'))):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        assert L.orc_sizeof_params() == C.sizeof(OrcParams)
        assert L.orc_sizeof_result() == C.sizeof(OrcResult)
        u8p = C.c_void_p
        L.orc_bgr2hls_full.argtypes = [u8p, C.c_int, C.c_int, C.c_long, C.c_int, u8p]
        L.orc_match_ccoeff.argtypes = [u8p, C.c_int, C.c_int, C.c_long, u8p, C.c_int, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]
        L.orc_build_dial_masks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, u8p]
        L.orc_inrange.argtypes = [u8p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, u8p]
        L.orc_close3.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_hls_inrange_close.argtypes = [u8p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p,
                                            C.c_void_p, u8p, u8p]
        L.orc_largest_contour.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(C.c_double), u8p]
        L.orc_angle_by_vector.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.orc_value_by_positions.argtypes = [C.c_void_p]
        L.orc_value_by_positions.restype = C.c_double
        L.orc_read_dials.argtypes = [u8p, C.POINTER(OrcParams), u8p, C.c_void_p, C.POINTER(OrcResult)]
        L.orc_process_crop.argtypes = [u8p, C.c_int, C.c_int, C.c_long, C.POINTER(OrcParams), u8p,
                                       u8p, C.c_void_p, C.POINTER(OrcResult)]
        L.orc_process_frames.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.POINTER(OrcParams), u8p, u8p, C.c_void_p,
                                         C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# audit switches of melf_oracle.c (ORC_OPT_*): value 0 is the restatement proper
OPTIONS = ['hls_variant', 'contour_tie', 'mean_form', 'hls_round', 'area_rule', 'no_hole_fill', 'erode_border',
           'l_integer', 'minmax_last', 'hue_g_first']


def set_option(name, value):
    L = lib()
    assert L.orc_option_count() == len(OPTIONS)
    assert L.orc_set_option(OPTIONS.index(name), int(value)) == 0


class Params:
    """Independent restatement of meterelf/_params.py:30-81 (fields the hot
    path needs).  PyYAML here is 6.x, so SafeLoader is given explicitly."""

    def __init__(self, filename):
        with open(filename, 'rt') as fp:
            data = yaml.load(fp, Loader=yaml.SafeLoader)
        base = os.path.dirname(filename)
        self.filename = filename
        mr = data['meter_rect']
        self.meter_rect = (tuple(mr['top_left']), tuple(mr['bottom_right']))
        self.dials_file = os.path.join(base, data['dials_template'])
        self.match_threshold = data['dials_template_match_threshold']
        (w, h) = data['dials_template_size']
        self.template_size = (h, w)
        self.hue_shift = data['hue_shift']
        nc, ncr = data['needle_color'], data['needle_color_range']
        self.needle_color = (nc['h'], nc['l'], nc['s'])
        self.needle_color_range = (ncr['h'], ncr['l'], ncr['s'])
        self.dials = data['needle_data']
        self.names = [d['name'] for d in self.dials]
        self.template = None

    def load_template(self):
        from PIL import Image
        if self.template is None:
            im = Image.open(self.dials_file)
            assert im.mode == 'L', im.mode
            t = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
            assert t.shape == self.template_size
            self.template = t
        return self.template

    def c_params(self):
        p = OrcParams()
        p.th, p.tw = self.template_size
        p.hue_shift = self.hue_shift
        p.ndials = len(self.dials)
        p.match_threshold = float(self.match_threshold)
        for i, d in enumerate(self.dials):
            cr = d['color_range']
            p.dial[i].cx, p.dial[i].cy = d['center']
            p.dial[i].range_h, p.dial[i].range_l, p.dial[i].range_s = cr['h'], cr['l'], cr['s']
            p.dial[i].negative_momentum = 1 if d['negative_momentum'] else 0
            p.dial[i].angle_of_zero = d['angle_of_zero']
        return p

    def name_order(self):
        order = sorted(range(len(self.names)), key=lambda i: self.names[i])
        return np.array(order, dtype=np.int32)

    def masks(self):
        th, tw = self.template_size
        n = len(self.dials)
        centers = np.array([d['center'] for d in self.dials], dtype=np.float64)
        diam = np.array([d['diameter'] for d in self.dials], dtype=np.int32)
        dist = np.array([d['dist_from_center'] for d in self.dials], dtype=np.int32)
        thick = np.array([d['circle_thickness'] for d in self.dials], dtype=np.int32)
        out = np.zeros((n, 2, th, tw), dtype=np.uint8)
        lib().orc_build_dial_masks(th, tw, n, _ptr(centers), _ptr(diam), _ptr(dist), _ptr(thick), _ptr(out))
        return out


def decode_bgr(filename):
    """cv2.imread(filename): 8-bit BGR, HWC (meterelf/_image.py:49).  Pillow's
    libjpeg-turbo (ISLOW IDCT, fancy upsampling) stands in for cv2's."""
    from PIL import Image
    try:
        im = Image.open(filename)
        im = im.convert('RGB')
    except Exception:
        return None
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8)[:, :, ::-1])


def crop_meter(img, params):
    ((x0, y0), (x1, y1)) = params.meter_rect
    return np.ascontiguousarray(img[y0:y1, x0:x1])


def bgr2hls(bgr, hue_shift):
    bgr = np.ascontiguousarray(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_bgr2hls_full(_ptr(bgr), h, w, w * 3, hue_shift, _ptr(out))
    return out


def match_ccoeff(img, tpl, want_map=False):
    img = np.ascontiguousarray(img)
    tpl = np.ascontiguousarray(tpl)
    (h, w), (th, tw) = img.shape, tpl.shape
    res = np.empty((h - th + 1, w - tw + 1), np.float32) if want_map else None
    mv, mx, my = C.c_float(), C.c_int(), C.c_int()
    lib().orc_match_ccoeff(_ptr(img), h, w, w, _ptr(tpl), th, tw, _ptr(res) if want_map else None,
                           C.byref(mv), C.byref(mx), C.byref(my))
    return (mv.value, mx.value, my.value, res)


def hls_inrange_close(bgr, hue_shift, lo, hi, want_l=False):
    bgr = np.ascontiguousarray(bgr)
    h, w, _ = bgr.shape
    lo = np.array(lo, np.int32)
    hi = np.array(hi, np.int32)
    out = np.empty((h, w), np.uint8)
    lpl = np.empty((h, w), np.uint8) if want_l else None
    lib().orc_hls_inrange_close(_ptr(bgr), h, w, w * 3, hue_shift, _ptr(lo), _ptr(hi), _ptr(out),
                                _ptr(lpl) if want_l else None)
    return (out, lpl) if want_l else out


def largest_contour(binimg):
    binimg = np.ascontiguousarray(binimg, dtype=np.uint8)
    h, w = binimg.shape
    area = C.c_double()
    filled = np.zeros((h, w), np.uint8)
    n = lib().orc_largest_contour(_ptr(binimg), h, w, C.byref(area), _ptr(filled))
    return n, area.value, filled


def angle_by_vector(x, y):
    out = C.c_double()
    ok = lib().orc_angle_by_vector(float(x), float(y), C.byref(out))
    return out.value if ok else None


def value_by_positions(r4321):
    a = np.array(r4321, np.float64)
    return lib().orc_value_by_positions(_ptr(a))


def read_dials(dials_hls, params):
    dials_hls = np.ascontiguousarray(dials_hls)
    res = OrcResult()
    res.failed_dial = -1
    cp = params.c_params()
    masks = params.masks()
    order = params.name_order()
    lib().orc_read_dials(_ptr(dials_hls), C.byref(cp), _ptr(masks), _ptr(order), C.byref(res))
    return res


def process_crop(crop_bgr, params):
    crop_bgr = np.ascontiguousarray(crop_bgr)
    h, w, _ = crop_bgr.shape
    res = OrcResult()
    cp = params.c_params()
    tpl = params.load_template()
    masks = params.masks()
    order = params.name_order()
    lib().orc_process_crop(_ptr(crop_bgr), h, w, w * 3, C.byref(cp), _ptr(tpl), _ptr(masks), _ptr(order),
                           C.byref(res))
    return res


def process_frames(frames, params):
    """frames: (N, H, W, 3) u8 BGR.  Returns a ctypes array of OrcResult."""
    frames = np.ascontiguousarray(frames)
    n, H, W, _ = frames.shape
    res = (OrcResult * n)()
    cp = params.c_params()
    tpl = params.load_template()
    masks = params.masks()
    order = params.name_order()
    ((x0, y0), (x1, y1)) = params.meter_rect
    lib().orc_process_frames(_ptr(frames), n, H, W, H * W * 3, x0, y0, x1, y1, C.byref(cp), _ptr(tpl),
                             _ptr(masks), _ptr(order), C.cast(res, C.c_void_p))
    return res


def error_message(res, params, filename=''):
    """meterelf/exceptions.py:21-32 get_message() as the CLI prints it."""
    if res.status == DIALS_NOT_FOUND:
        return 'Dials not found (match val = {})'.format(float(res.match_val))
    if res.status == NEEDLE_CONTOURS_NOT_FOUND:
        return 'Cannot find needle contours of a dial (dial = {})'.format(params.names[res.failed_dial])
    if res.status == ANGLE_UNDETERMINED:
        bad = [params.names[i] for i in range(len(params.names)) if res.unreadable_mask >> i & 1]
        return 'Cannot determine angle of a dial (unreadable dials = {})'.format(', '.join(bad))
    return None


def output_line(filename, res, params):
    """meterelf/_main.py:16-22."""
    value = res.value if res.status == OK and len(params.names) == 4 else None
    value_str = '{:07.3f}'.format(value) if value else ''
    msg = error_message(res, params, filename)
    error_str = 'UNKNOWN {}'.format(msg) if msg else ''
    return '{}: {}{}'.format(filename, value_str, error_str)


def run_file(filename, params, display_name=None):
    img = decode_bgr(filename)
    name = display_name if display_name is not None else filename
    if img is None:
        return '{}: UNKNOWN Unable to load image'.format(name), None
    res = process_crop(crop_meter(img, params), params)
    return output_line(name, res, params), res
