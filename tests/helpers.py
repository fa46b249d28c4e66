"""Test-only glue: lets the CPU oracle stand in for the GPU context so that the
product's HOST logic (API, chunking, error conversion, sharding) can be
exercised without a GPU.  Never imported by meterelf_amd."""
import numpy as np

from meterelf_amd import _hip
from oracle import pyoracle as po


def orc_to_records(ores, n):
    out = np.zeros(n, _hip.RESULT_DTYPE)
    for i in range(n):
        o = ores[i]
        out[i]['status'] = o.status
        out[i]['match_x'], out[i]['match_y'] = o.match_x, o.match_y
        out[i]['failed_dial'] = o.failed_dial
        out[i]['unreadable_mask'] = o.unreadable_mask
        out[i]['match_val'] = o.match_val
        out[i]['pos'][:] = list(o.pos)
        out[i]['angle'][:] = list(o.angle)
        out[i]['value'] = o.value
    return out


class OracleReader:
    """Same surface as meterelf_amd.MeterReader, computed by the oracle."""

    def __init__(self, params, device=0, blob=None):
        self.params = params
        self.dial_names = params.dial_names
        self.op = po.Params(params_file_of(params))

    def close(self):
        pass

    def read_frames(self, frames):
        return orc_to_records(po.process_frames(frames, self.op), len(frames))

    def read_jpeg_files(self, files):
        """No GPU decoder here: every file goes back to the caller's host-decode branch."""
        return [None] * len(files)

    def read_jpeg_paths(self, paths):
        return [None] * len(paths)

    def read_many(self, images, cropped=None):
        return [self.read_frames(img[None])[0] for img in images]


def params_file_of(params):
    import os
    return os.path.join(os.path.dirname(params.dials_file), 'params.yml')


def hip_runtime():
    """The HIP runtime instance libmeterelf_hip.so is bound to, for tests that allocate device memory themselves.
    ctypes.CDLL('libamdhip64.so') is not good enough: once torch has been imported the process can hold two copies
    (torch ships its own and loads it by path), and device memory of one is unknown to the other (hipMalloc fails with
    'no device' or the pointers are foreign).  The library is always loaded before torch in this suite, so its copy is
    the mapped one that does not live under torch/."""
    import ctypes as C

    from meterelf_amd import _hip
    _hip.lib()
    paths = sorted({line.split()[-1] for line in open('/proc/self/maps') if 'libamdhip64' in line})
    ours = [p for p in paths if '/torch/' not in p] or paths
    return C.CDLL(ours[0])


def libpng_rgb_to_gray(path):
    """The grey image libpng makes of a colour / palette PNG when asked the way OpenCV 3.4's PNG reader asks
    (modules/imgcodecs/src/grfmt_png.cpp: png_set_palette_to_rgb for palettes, png_set_strip_alpha, then
    png_set_rgb_to_gray(png, 1, 0.299, 0.587)), through ctypes on the system's libpng16.  None when there is no libpng16 to
    load (the formula assertion beside it still stands).  Test infrastructure: the expected vector comes from the library
    the reference's dependency links, not from a restated formula."""
    import ctypes as C
    try:
        png = C.CDLL('libpng16.so.16')
    except OSError:
        return None
    libc = C.CDLL(None)
    vp = C.c_void_p
    libc.fopen.restype = vp
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [vp]
    png.png_get_libpng_ver.restype = C.c_char_p
    png.png_get_libpng_ver.argtypes = [vp]
    png.png_create_read_struct.restype = vp
    png.png_create_read_struct.argtypes = [C.c_char_p, vp, vp, vp]
    png.png_create_info_struct.restype = vp
    png.png_create_info_struct.argtypes = [vp]
    for (name, args) in (('png_init_io', [vp, vp]), ('png_read_info', [vp, vp]), ('png_read_update_info', [vp, vp]),
                         ('png_set_palette_to_rgb', [vp]), ('png_set_strip_alpha', [vp]), ('png_read_image', [vp, vp]),
                         ('png_set_rgb_to_gray', [vp, C.c_int, C.c_double, C.c_double])):
        getattr(png, name).restype = None
        getattr(png, name).argtypes = args
    for (name, rt) in (('png_get_image_width', C.c_uint32), ('png_get_image_height', C.c_uint32), ('png_get_color_type', C.c_ubyte),
                       ('png_get_rowbytes', C.c_size_t)):
        getattr(png, name).restype = rt
        getattr(png, name).argtypes = [vp, vp]
    fp = libc.fopen(path.encode(), b'rb')
    assert fp, path
    try:
        p = png.png_create_read_struct(png.png_get_libpng_ver(None), None, None, None)
        info = png.png_create_info_struct(p)
        assert p and info
        png.png_init_io(p, fp)
        png.png_read_info(p, info)
        (w, h, ct) = (png.png_get_image_width(p, info), png.png_get_image_height(p, info), png.png_get_color_type(p, info))
        if ct == 3:
            png.png_set_palette_to_rgb(p)
        if ct & 4:
            png.png_set_strip_alpha(p)
        if ct & 2 or ct == 3:
            png.png_set_rgb_to_gray(p, 1, 0.299, 0.587)
        png.png_read_update_info(p, info)
        assert png.png_get_rowbytes(p, info) == w
        out = np.zeros((h, w), np.uint8)
        rows = (vp * h)(*[out.ctypes.data + i * w for i in range(h)])
        png.png_read_image(p, rows)
    finally:
        libc.fclose(fp)
    return out
