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
