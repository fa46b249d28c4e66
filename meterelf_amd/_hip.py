"""ctypes binding of libmeterelf_hip.so (C ABI: include/meterelf_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or no
MI355X is visible, everything here raises.
"""
import ctypes as C
import os
import sys

import numpy as np

# Streams overlap on the GPU only if they sit on different hardware queues; the HIP runtime spreads all of a process's
# streams over GPU_MAX_HW_QUEUES = 4 by default.  Eight keeps the context's two lanes (two caller streams) apart from each
# other and from the copy stream in practice.  Read when HIP initialises: harmless if that has already happened.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MELF_LIB_PATH') or os.path.join(_PKG, 'libmeterelf_hip.so')  # override: A/B builds

MAX_DIALS = 8
ABI_VERSION = 3

FRAME_OK = 0
FRAME_DIALS_NOT_FOUND = 1
FRAME_NEEDLE_CONTOURS_NOT_FOUND = 2
FRAME_ANGLE_UNDETERMINED = 3

K_LPLANE, K_MATCH, K_DIALS, K_FUSED_MASK, K_HLS, K_JPEG_HUFF, K_JPEG_IDCT, K_JPEG_COLOR, K_STREAM_PROBE, K_COUNT = range(10)
JPEG_OK, JPEG_UNSUPPORTED, JPEG_CORRUPT, JPEG_SIZE_MISMATCH, JPEG_UNREADABLE = 0, 1, 2, 3, 4
FILES_IN_FLIGHT_MAX = 3   # MELF_FILES_IN_FLIGHT_MAX (include/meterelf_hip.h); tests compare with melf_jpeg_files_in_flight_max()


class MelfDial(C.Structure):
    _fields_ = [('cx', C.c_double), ('cy', C.c_double), ('angle_of_zero', C.c_double),
                ('range_h', C.c_int32), ('range_l', C.c_int32), ('range_s', C.c_int32),
                ('negative_momentum', C.c_int32), ('diameter', C.c_int32),
                ('dist_from_center', C.c_int32), ('circle_thickness', C.c_int32),
                ('reserved', C.c_int32)]


class MelfParams(C.Structure):
    _fields_ = [('abi_version', C.c_int32),
                ('rect_x0', C.c_int32), ('rect_y0', C.c_int32), ('rect_x1', C.c_int32), ('rect_y1', C.c_int32),
                ('th', C.c_int32), ('tw', C.c_int32), ('hue_shift', C.c_int32), ('ndials', C.c_int32),
                ('needle_lo', C.c_int32 * 3), ('needle_hi', C.c_int32 * 3),
                ('name_order', C.c_int32 * MAX_DIALS), ('reserved', C.c_int32),
                ('match_threshold', C.c_double),
                ('dial', MelfDial * MAX_DIALS)]


class MelfResult(C.Structure):
    _fields_ = [('status', C.c_int32), ('match_x', C.c_int32), ('match_y', C.c_int32),
                ('failed_dial', C.c_int32), ('unreadable_mask', C.c_uint32), ('match_val', C.c_float),
                ('pos', C.c_double * MAX_DIALS), ('angle', C.c_double * MAX_DIALS), ('value', C.c_double)]


class MelfMatchInfo(C.Structure):
    _fields_ = [('kernel', C.c_int32), ('n', C.c_int32), ('rows', C.c_int32), ('cols', C.c_int32), ('groups', C.c_int32),
                ('waves', C.c_int32), ('rows_per_wave', C.c_int32), ('full_waves', C.c_int32), ('pair_waves', C.c_int32),
                ('tiles', C.c_int32), ('reserved', C.c_int32 * 6)]


MATCH_KERNEL_NAMES = ('dot4', 'mfma', 'gen')

RESULT_DTYPE = np.dtype([('status', '<i4'), ('match_x', '<i4'), ('match_y', '<i4'), ('failed_dial', '<i4'),
                         ('unreadable_mask', '<u4'), ('match_val', '<f4'),
                         ('pos', '<f8', (MAX_DIALS,)), ('angle', '<f8', (MAX_DIALS,)), ('value', '<f8')])
assert RESULT_DTYPE.itemsize == C.sizeof(MelfResult)


class HipError(RuntimeError):
    pass


# every symbol include/meterelf_hip.h declares for the product library; DIAG_EXPORTS: what its `#ifdef MELF_DIAG` part adds (the
# diagnostic build, make -C meterelf_amd/csrc diag, loaded through MELF_LIB_PATH)
DIAG_EXPORTS = ['melf_stream_probe_dev']
EXPORTS = [
    'melf_last_error', 'melf_abi_version', 'melf_device_count', 'melf_build_dial_masks',
    'melf_blob_size', 'melf_blob_pack', 'melf_blob_params', 'melf_ctx_create', 'melf_ctx_create_bcast', 'melf_ctx_destroy',
    'melf_ctx_params', 'melf_ctx_sync', 'melf_ctx_get_masks', 'melf_process_batch', 'melf_process_batch_dev', 'melf_process_stream_dev',
    'melf_bgr2hls', 'melf_hls_inrange_close', 'melf_hls_inrange_close_dev', 'melf_match_ccoeff',
    'melf_read_dials', 'melf_aligned_average', 'melf_inrange', 'melf_ctx_fused_table_ties', 'melf_ctx_set_frames_resident', 'melf_ctx_last_match', 'melf_match_layout_query', 'melf_match_gen_plan_query', 'melf_ctx_set_profiling', 'melf_ctx_timings', 'melf_kernel_name',
    'melf_jpeg_probe', 'melf_jpeg_probe_batch', 'melf_jpeg_decode_batch', 'melf_jpeg_clean_segment', 'melf_jpeg_process_batch',
    'melf_jpeg_process_files', 'melf_jpeg_process_files_begin', 'melf_jpeg_process_files_end', 'melf_jpeg_files_in_flight_max', 'melf_ctx_files_stats', 'melf_files_open_probe',
]

_lib = None


def lib():
    """Loads the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(
            'libmeterelf_hip.so is not built (run `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C meterelf_amd/csrc`); meterelf_amd has no CPU fallback')
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.melf_last_error.restype = C.c_char_p
    L.melf_kernel_name.restype = C.c_char_p
    L.melf_kernel_name.argtypes = [C.c_int]
    L.melf_device_count.argtypes = [C.POINTER(C.c_int)]
    L.melf_build_dial_masks.argtypes = [C.POINTER(MelfParams), vp]
    L.melf_blob_size.restype = C.c_size_t
    L.melf_blob_size.argtypes = [C.POINTER(MelfParams)]
    L.melf_blob_pack.argtypes = [C.POINTER(MelfParams), vp, vp, C.c_size_t]
    L.melf_blob_params.argtypes = [vp, C.c_size_t, C.POINTER(MelfParams)]
    L.melf_ctx_create.argtypes = [C.c_int, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.melf_ctx_create_bcast.argtypes = [C.POINTER(C.c_int), C.c_int, vp, C.c_size_t, C.POINTER(vp)]
    L.melf_ctx_destroy.argtypes = [vp]
    L.melf_ctx_destroy.restype = None
    L.melf_ctx_params.argtypes = [vp, C.POINTER(MelfParams)]
    L.melf_ctx_get_masks.argtypes = [vp, vp]
    L.melf_ctx_sync.argtypes = [vp]
    L.melf_process_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, vp]
    L.melf_process_batch_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, vp]
    L.melf_process_stream_dev.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, vp]
    L.melf_bgr2hls.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp]
    L.melf_hls_inrange_close.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.melf_hls_inrange_close_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    if hasattr(L, 'melf_stream_probe_dev'):   # the diagnostic build only
        L.melf_stream_probe_dev.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
    L.melf_match_ccoeff.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.melf_read_dials.argtypes = [vp, vp, C.c_int, vp]
    L.melf_aligned_average.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_int, C.c_int, vp]
    L.melf_inrange.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.melf_ctx_fused_table_ties.argtypes = [vp, C.POINTER(C.c_int)]
    L.melf_ctx_set_profiling.argtypes = [vp, C.c_int]
    L.melf_ctx_last_match.argtypes = [vp, C.POINTER(MelfMatchInfo)]
    L.melf_match_layout_query.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MelfMatchInfo)]
    L.melf_match_gen_plan_query.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MelfMatchInfo), vp, C.c_int,
                                            C.POINTER(C.c_int32)]
    L.melf_ctx_set_frames_resident.argtypes = [vp, C.c_int]
    L.melf_ctx_timings.argtypes = [vp, vp, vp]
    i32p = C.POINTER(C.c_int32)
    L.melf_jpeg_probe.argtypes = [vp, C.c_size_t, i32p, i32p, i32p]
    L.melf_jpeg_probe_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.melf_jpeg_decode_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
    L.melf_jpeg_process_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.melf_jpeg_process_files.argtypes = [vp, vp, C.c_int, i32p, i32p, vp, vp]
    L.melf_jpeg_process_files_begin.argtypes = [vp, vp, C.c_int, i32p, i32p, vp, vp]
    L.melf_jpeg_process_files_end.argtypes = [vp]
    L.melf_jpeg_files_in_flight_max.argtypes = []
    L.melf_ctx_files_stats.argtypes = [vp, vp, C.c_int]
    L.melf_files_open_probe.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    if L.melf_abi_version() != ABI_VERSION:
        raise HipError('libmeterelf_hip.so ABI version mismatch')
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise HipError('libmeterelf_hip: %s (code %d)' % (lib().melf_last_error().decode(), rc))


def device_count():
    n = C.c_int(0)
    rc = lib().melf_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def jpeg_probe(data):
    """(H, W, supported, reason) of a JPEG file's bytes; header parse only, no GPU."""
    L = lib()
    H, W, ok = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    if isinstance(data, np.ndarray):
        p = _ptr(data)
    else:
        data = data if isinstance(data, bytes) else bytes(data)
        p = C.cast(C.c_char_p(data), C.c_void_p)  # no copy: the bytes object outlives the call
    check(L.melf_jpeg_probe(p, len(data), C.byref(H), C.byref(W), C.byref(ok)))
    return H.value, W.value, bool(ok.value), L.melf_last_error().decode()


def jpeg_probe_batch(files):
    """Header check of many files' bytes in one call: (H, W, supported) int32 arrays."""
    n = len(files)
    (H, W, ok) = (np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32))
    if n:
        (ptrs, sizes, keep) = _file_table(files)
        check(lib().melf_jpeg_probe_batch(ptrs, sizes, n, _ptr(H), _ptr(W), _ptr(ok)))
    return H, W, ok


def _file_table(files):
    """list of bytes objects -> (pointer array, size array, keep-alive).  One ctypes array construction for the whole list
    (the per-file cast of a c_char_p cost 0.9 ms per 1024 files, a quarter of a melf_jpeg_process_batch call)."""
    n = len(files)
    keep = [f if isinstance(f, bytes) else bytes(f) for f in files]
    ptrs = (C.c_char_p * n)(*keep)          # holds references to the bytes objects; no copies
    sizes = np.fromiter(map(len, keep), dtype=np.uint64, count=n)
    return ptrs, _ptr(sizes), (keep, sizes)


file_table = _file_table


def _path_table(paths):
    """list of file names (str or bytes) -> (char** as a numpy array of addresses, keep-alive).  One join + one encode for the
    whole list instead of an os.fsencode and a ctypes conversion per name (0.45 -> 0.05 ms per 1024 names: the Python side
    of get_meter_values is one thread, and its per-chunk cost is what long lists wait for)."""
    n = len(paths)
    if n == 0:
        return np.zeros(1, np.uint64), None
    if all(type(p) is str for p in paths):
        blob = ('\0'.join(paths) + '\0').encode(sys.getfilesystemencoding(), 'surrogateescape')   # = os.fsencode, for all at once
    else:
        blob = b'\0'.join(os.fsencode(p) for p in paths) + b'\0'
    buf = np.frombuffer(blob, np.uint8)
    ends = np.flatnonzero(buf == 0)
    if len(ends) != n:   # a name with an embedded NUL: a C string would silently end there ('good.jpg\0x' would open good.jpg)
        raise ValueError('embedded null byte')   # what open() raises for such a name
    addr = np.empty(n, np.uint64)
    addr[0] = 0
    addr[1:] = ends[:-1] + 1
    addr += np.uint64(buf.ctypes.data)
    return addr, (blob, buf)


def files_open_probe(paths, device=0):
    """open() + close() of every path on the library's I/O pool: (milliseconds, threads).  Measurement aid."""
    (addr, keep) = _path_table(paths)
    ms = C.c_double(0.0)
    th = C.c_int(0)
    check(lib().melf_files_open_probe(C.c_void_p(addr.ctypes.data), len(paths), device, C.byref(ms), C.byref(th)))
    del keep
    return ms.value, th.value


def pack_blob(cparams, template):
    """params + template -> calibration blob (numpy uint8), masks built inside."""
    L = lib()
    template = np.ascontiguousarray(template, dtype=np.uint8)
    assert template.shape == (cparams.th, cparams.tw), (template.shape, cparams.th, cparams.tw)
    size = L.melf_blob_size(C.byref(cparams))
    if size == 0:
        raise HipError('libmeterelf_hip: %s' % L.melf_last_error().decode())
    blob = np.zeros(size, np.uint8)
    check(L.melf_blob_pack(C.byref(cparams), _ptr(template), _ptr(blob), size))
    return blob


def blob_params(blob):
    p = MelfParams()
    check(lib().melf_blob_params(_ptr(blob), blob.nbytes, C.byref(p)))
    return p


def build_dial_masks(cparams):
    out = np.zeros((cparams.ndials, 2, cparams.th, cparams.tw), np.uint8)
    check(lib().melf_build_dial_masks(C.byref(cparams), _ptr(out)))
    return out


GEN_TASK_DTYPE = np.dtype([(k, '<i4') for k in ('y0', 'rows', 'rows_computed', 'xb0', 'nxb', 'tile', 'slice', 'nslices', 'k_lo', 'k_hi',
                                                  'lds_bytes', 'reserved')])


def _match_info_dict(mi, gen_plan=False):
    d = {k: getattr(mi, k) for (k, _t) in MelfMatchInfo._fields_ if k != 'reserved'}
    d['kernel'] = MATCH_KERNEL_NAMES[mi.kernel]
    d['layout'] = ('rb%d%s%s' % (mi.rows_per_wave, '+pairs' if mi.pair_waves else '', '/k%d' % mi.reserved[2] if mi.reserved[2] > 1 else '')) if mi.kernel == 1 and not gen_plan else None
    if mi.kernel == 2 or gen_plan:   # the general kernel's plan: tile shape, slices, remainder columns
        (d['nd'], d['rows_pad'], d['blocks_per_tile'], d['slices'], d['v_columns'], d['v_blocks']) = tuple(mi.reserved)
        d['layout'] = 'r%dx%d/%d%s' % (mi.rows_per_wave, d['blocks_per_tile'], d['slices'], '+v%d' % d['v_columns'] if d['v_columns'] else '')
    else:
        d['th_pad'], d['rows_pad'], d['k_slices'] = mi.reserved[0], mi.reserved[1], max(1, mi.reserved[2])
    return d


def match_gen_plan_query(th, tw, rows, cols, n):
    """The general matrix-core kernel's plan for this shape and batch size (host logic, no GPU needed): (info dict -- 'kernel' is what
    DEFAULT dispatch launches for the shape --, structured array of one frame group's wave tasks)."""
    mi = MelfMatchInfo()
    nt = C.c_int32(0)
    check(lib().melf_match_gen_plan_query(th, tw, rows, cols, n, C.byref(mi), None, 0, C.byref(nt)))
    tasks = np.zeros(nt.value, GEN_TASK_DTYPE)
    check(lib().melf_match_gen_plan_query(th, tw, rows, cols, n, C.byref(mi), _ptr(tasks), nt.value, C.byref(nt)))
    d = _match_info_dict(mi, gen_plan=True)
    d['default_kernel'] = d.pop('kernel')
    return d, tasks


def match_layout_query(th, tw, rows, cols, n):
    """The tuned matrix-core kernel's wave layout for this shape and batch size (host logic, no GPU needed)."""
    mi = MelfMatchInfo()
    check(lib().melf_match_layout_query(th, tw, rows, cols, n, C.byref(mi)))
    return _match_info_dict(mi)


class Context:
    """One melf_ctx = one GPU's resident calibration state + workspaces."""

    def __init__(self, blob, device=0, blob_device_ptr=None):
        L = lib()
        self._h = C.c_void_p()
        self._L = L
        if blob_device_ptr is not None:
            check(L.melf_ctx_create(device, C.c_void_p(blob_device_ptr), int(blob.nbytes), 1, C.byref(self._h)))
        else:
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
            check(L.melf_ctx_create(device, _ptr(blob), blob.nbytes, 0, C.byref(self._h)))
        self.device = device
        self.params = MelfParams()
        check(L.melf_ctx_params(self._h, C.byref(self.params)))

    @classmethod
    def create_bcast(cls, blob, devices):
        """One context per listed GPU of THIS process, the blob sent to devices[0] and broadcast from there by RCCL
        (melf_ctx_create_bcast).  Devices must be distinct.  Returns the contexts in the order of `devices`."""
        L = lib()
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        n = len(devices)
        devs = (C.c_int * n)(*[int(d) for d in devices])
        handles = (C.c_void_p * n)()
        check(L.melf_ctx_create_bcast(devs, n, _ptr(blob), blob.nbytes, handles))
        out = []
        for (d, h) in zip(devices, handles):
            c = cls.__new__(cls)
            c._h = C.c_void_p(h)
            c._L = L
            c.device = int(d)
            c.params = MelfParams()
            check(L.melf_ctx_params(c._h, C.byref(c.params)))
            out.append(c)
        return out

    def sync(self):
        """Waits for the context's work on every caller stream and forgets the streams (call before destroying one)."""
        check(self._L.melf_ctx_sync(self._h))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._L.melf_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- whole path ---
    def process_batch(self, frames):
        """frames: (N, H, W, 3) uint8 BGR host array -> structured array of records."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        assert frames.ndim == 4 and frames.shape[3] == 3, frames.shape
        n, H, W, _ = frames.shape
        out = np.zeros(n, RESULT_DTYPE)
        if n:
            check(self._L.melf_process_batch(self._h, _ptr(frames), n, H, W, H * W * 3, _ptr(out)))
        return out

    def process_batch_dev(self, d_frames_ptr, n, H, W, frame_stride=None, d_results_ptr=None, want_host=True,
                          stream=None):
        """Frames already in HBM (device pointer as int).  Returns records when want_host."""
        out = np.zeros(n, RESULT_DTYPE) if want_host else None
        check(self._L.melf_process_batch_dev(
            self._h, C.c_void_p(d_frames_ptr), n, H, W, frame_stride or H * W * 3,
            C.c_void_p(d_results_ptr) if d_results_ptr else None,
            _ptr(out) if want_host else None, C.c_void_p(stream) if stream else None))
        return out

    # --- stages ---
    def process_stream_dev(self, d_frames_ptr, nbatches, batch_stride, n, H, W, d_results_ptr, results_stride, frame_stride=None,
                           stream=None):
        """nbatches batches of n device-resident frames in one call; consecutive batches overlap on two lanes."""
        check(self._L.melf_process_stream_dev(self._h, C.c_void_p(d_frames_ptr), nbatches, batch_stride, n, H, W,
                                              frame_stride or H * W * 3, C.c_void_p(d_results_ptr), results_stride,
                                              C.c_void_p(stream) if stream else None))

    def bgr2hls(self, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        rows, cols, _ = bgr.shape
        out = np.empty((rows, cols, 3), np.uint8)
        check(self._L.melf_bgr2hls(self._h, _ptr(bgr), rows, cols, cols * 3, _ptr(out)))
        return out

    def hls_inrange_close(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, H, W, _ = frames.shape
        out = np.empty((n, H, W), np.uint8)
        check(self._L.melf_hls_inrange_close(self._h, _ptr(frames), n, H, W, _ptr(out)))
        return out

    def hls_inrange_close_dev(self, d_frames_ptr, n, H, W, d_masks_ptr, stream=None):
        check(self._L.melf_hls_inrange_close_dev(self._h, C.c_void_p(d_frames_ptr), n, H, W,
                                                  C.c_void_p(d_masks_ptr), C.c_void_p(stream) if stream else None))

    def stream_probe_dev(self, d_in_ptr, in_bytes, d_out_ptr, chunks_per_block=0, stream=None):
        if not hasattr(self._L, 'melf_stream_probe_dev'):
            raise HipError('melf_stream_probe_dev is part of the diagnostic build only (make -C meterelf_amd/csrc diag; MELF_LIB_PATH)')
        """Measurement aid: one bare 3:1 stream launch over device buffers (include/meterelf_hip.h); d_out is overwritten."""
        check(self._L.melf_stream_probe_dev(self._h, C.c_void_p(d_in_ptr), C.c_size_t(in_bytes), C.c_void_p(d_out_ptr),
                                             int(chunks_per_block), C.c_void_p(stream) if stream else None))

    def match_ccoeff(self, images, want_map=False):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n, rows, cols = images.shape
        mv = np.zeros(n, np.float32)
        mx = np.zeros(n, np.int32)
        my = np.zeros(n, np.int32)
        rmap = None
        if want_map:
            rmap = np.zeros((n, rows - self.params.th + 1, cols - self.params.tw + 1), np.float32)
        check(self._L.melf_match_ccoeff(self._h, _ptr(images), n, rows, cols, _ptr(mv), _ptr(mx), _ptr(my),
                                        _ptr(rmap) if want_map else None))
        return mv, mx, my, rmap

    def read_dials(self, dials_hls):
        dials_hls = np.ascontiguousarray(dials_hls, dtype=np.uint8)
        n = dials_hls.shape[0]
        assert dials_hls.shape[1:] == (self.params.th, self.params.tw, 3), dials_hls.shape
        out = np.zeros(n, RESULT_DTYPE)
        check(self._L.melf_read_dials(self._h, _ptr(dials_hls), n, _ptr(out)))
        return out

    def masks(self):
        p = self.params
        out = np.zeros((p.ndials, 2, p.th, p.tw), np.uint8)
        check(self._L.melf_ctx_get_masks(self._h, _ptr(out)))
        return out

    # --- calibration stages ---
    def aligned_average(self, frames, match_x, match_y, align_x, align_y):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        (n, H, W, _) = frames.shape
        p = self.params
        (rows, cols) = (min(p.rect_y1, H) - min(p.rect_y0, H), min(p.rect_x1, W) - min(p.rect_x0, W))
        mx = np.ascontiguousarray(match_x, dtype=np.int32)
        my = np.ascontiguousarray(match_y, dtype=np.int32)
        out = np.empty((rows, cols, 3), np.uint8)
        check(self._L.melf_aligned_average(self._h, _ptr(frames), n, H, W, H * W * 3, _ptr(mx), _ptr(my),
                                           int(align_x), int(align_y), _ptr(out)))
        return out

    def inrange(self, img, lo, hi):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        (rows, cols, _) = img.shape
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        out = np.empty((rows, cols), np.uint8)
        check(self._L.melf_inrange(self._h, _ptr(img), rows, cols, _ptr(lo), _ptr(hi), _ptr(out)))
        return out

    def fused_table_ties(self):
        n = C.c_int(0)
        check(self._L.melf_ctx_fused_table_ties(self._h, C.byref(n)))
        return n.value

    # --- measurement ---
    def jpeg_decode(self, files, H, W):
        """cv2.imread for a batch of JPEG files' bytes: (frames n x H x W x 3 BGR u8, status n)."""
        n = len(files)
        out = np.zeros((n, H, W, 3), np.uint8)
        status = np.zeros(n, np.int32)
        if n:
            (ptrs, sizes, keep) = _file_table(files)
            check(self._L.melf_jpeg_decode_batch(self._h, ptrs, sizes, n, H, W, _ptr(out), 0, _ptr(status)))
        return out, status

    def jpeg_decode_dev(self, files, H, W, d_frames_ptr):
        """Same, into device memory (n*H*W*3 bytes at d_frames_ptr); returns the status array."""
        n = len(files)
        status = np.zeros(n, np.int32)
        if n:
            (ptrs, sizes, keep) = _file_table(files)
            check(self._L.melf_jpeg_decode_batch(self._h, ptrs, sizes, n, H, W, C.c_void_p(d_frames_ptr), 1, _ptr(status)))
        return status

    def jpeg_process_batch(self, files, H, W, table=None):
        """JPEG bytes -> (result records, decode status); decode and reading both on the GPU.  table: file_table(files) made
        earlier (a caller that sends the same list again -- a benchmark -- then pays for the call alone, as a compiled host
        would, not for 1024 ctypes conversions)."""
        n = len(files)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        status = np.zeros(n, np.int32)
        if n:
            (ptrs, sizes, keep) = table if table is not None else _file_table(files)
            check(self._L.melf_jpeg_process_batch(self._h, ptrs, sizes, n, H, W, _ptr(out), _ptr(status)))
        return out, status

    def jpeg_process_files(self, paths):
        """File names -> (records, status, (H, W) of the batch): files are read, decoded and read out inside the
        library; status 3 = another frame size (call again with those), 1 / 2 / 4 = not for the GPU decoder."""
        n = len(paths)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        status = np.zeros(n, np.int32)
        (H, W) = (C.c_int32(0), C.c_int32(0))
        if n:
            (addr, keep) = _path_table(paths)
            check(self._L.melf_jpeg_process_files(self._h, _ptr(addr), n, C.byref(H), C.byref(W), _ptr(out), _ptr(status)))
        return out, status, (H.value, W.value)

    def jpeg_process_files_begin(self, paths):
        """Starts jpeg_process_files(paths) on a thread of the library and returns at once; jpeg_process_files_end()
        waits for the oldest call begun.  Up to FILES_IN_FLIGHT_MAX calls in flight per context (one reading its files, one
        preparing and enqueueing, one waiting for its kernels), no other call on the context in between."""
        n = len(paths)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        status = np.zeros(n, np.int32)
        hw = (C.c_int32(0), C.c_int32(0))
        (addr, keep) = _path_table(paths)
        check(self._L.melf_jpeg_process_files_begin(self._h, _ptr(addr), n, C.byref(hw[0]), C.byref(hw[1]), _ptr(out), _ptr(status)))
        if getattr(self, '_files_pending', None) is None:
            self._files_pending = []
        self._files_pending.append((out, status, hw, addr, keep))  # everything the library points into, alive until its _end

    def jpeg_process_files_end(self):
        (out, status, hw, _arr, _enc) = self._files_pending.pop(0)  # the library forgets the call whatever it returns
        check(self._L.melf_jpeg_process_files_end(self._h))
        return out, status, (hw[0].value, hw[1].value)

    def files_in_flight(self):
        return len(getattr(self, '_files_pending', None) or ())

    def set_frames_resident(self, on):
        """Promise that the frames of every process_batch_dev call are complete in HBM when the call is made: a call's prep
        kernels then run under the previous call's dials kernel (melf_ctx_set_frames_resident)."""
        check(self._L.melf_ctx_set_frames_resident(self._h, int(bool(on))))

    def last_match(self):
        """Kernel and wave layout of the most recent template-match launch: dict with 'kernel' ('dot4' / 'mfma' / 'gen'),
        'layout' (tuned kernel: 'rb<rows per wave>' + '+pairs' when pairs of waves share a map row), and the raw fields."""
        mi = MelfMatchInfo()
        check(self._L.melf_ctx_last_match(self._h, C.byref(mi)))
        return _match_info_dict(mi)

    def set_profiling(self, on):
        check(self._L.melf_ctx_set_profiling(self._h, int(on)))  # False/0 off, True/1 every kernel, 2 only k_match

    def files_stats(self, reset=True):
        """Host-time breakdown of the file-name calls since the last reset (include/meterelf_hip.h: melf_ctx_files_stats)."""
        out = np.zeros(10, np.float64)
        check(self._L.melf_ctx_files_stats(self._h, _ptr(out), 1 if reset else 0))
        keys = ('calls', 'files', 'ms_read', 'ms_turn_wait', 'ms_enqueue', 'ms_gpu_wait', 'io_threads', 'host_threads', 'cores', 'devices_in_process')
        return dict(zip(keys, (float(v) for v in out)))

    def timings(self):
        ms = np.zeros(K_COUNT, np.float64)
        cnt = np.zeros(K_COUNT, np.int64)
        check(self._L.melf_ctx_timings(self._h, _ptr(ms), _ptr(cnt)))
        return {self._L.melf_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(K_COUNT)}
