"""CPU-side tests (-m "not gpu"): the C ABI loads and exports every symbol the
header declares, struct layouts agree between C and ctypes, and the host logic
(params loader, dial masks, blob, error conversion, API/CLI formatting) matches
the oracle / the reference's goldens.  No GPU compute is called here."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

from meterelf_amd import _api, _engine, _hip, _main, _params, exceptions
from meterelf_amd._types import HlsColor, Rect
from oracle import pyoracle as po
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
HEADER = os.path.join(ROOT, 'include', 'meterelf_hip.h')


def test_library_exports_every_declared_symbol():
    with open(HEADER) as fp:
        text = re.sub(r'/\*.*?\*/', '', fp.read(), flags=re.S)
    # the `#ifdef MELF_DIAG` part of the header belongs to the diagnostic build (make diag): the product library must NOT export it
    diag_text = ''.join(re.findall(r'#ifdef MELF_DIAG(.*?)#endif', text, flags=re.S))
    diag_declared = sorted(set(re.findall(r'\b(melf_[a-z0-9_]+)\s*\(', diag_text)))
    text = re.sub(r'#ifdef MELF_DIAG.*?#endif', '', text, flags=re.S)
    declared = sorted(set(re.findall(r'\b(melf_[a-z0-9_]+)\s*\(', text)))
    assert len(declared) >= 20
    L = _hip.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_hip.EXPORTS) == declared and sorted(_hip.DIAG_EXPORTS) == diag_declared
    if 'MELF_LIB_PATH' not in os.environ:
        for name in diag_declared:
            assert not hasattr(L, name), name
        # ... and nothing else either: the dynamic symbol table is the header's list (plus nothing of the measurement scaffolding)
        import subprocess
        nm = subprocess.run(['nm', '-D', '--defined-only', _hip.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
        exported = sorted(ln.split()[-1] for ln in nm.splitlines() if ' T ' in ln and ln.split()[-1].startswith('melf_'))
        assert exported == declared, sorted(set(exported) ^ set(declared))
    # constants the Python side mirrors from the header
    assert L.melf_jpeg_files_in_flight_max() == _hip.FILES_IN_FLIGHT_MAX == int(re.search(r'#define MELF_FILES_IN_FLIGHT_MAX (\d+)', text).group(1))


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "meterelf_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(melf_params), sizeof(melf_result),'
                   'sizeof(melf_dial), offsetof(melf_params, match_threshold), offsetof(melf_params, dial),'
                   'offsetof(melf_result, pos), offsetof(melf_result, value), sizeof(melf_match_info),'
                   'offsetof(melf_match_info, tiles));return 0;}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(_hip.MelfParams), C.sizeof(_hip.MelfResult), C.sizeof(_hip.MelfDial),
           _hip.MelfParams.match_threshold.offset, _hip.MelfParams.dial.offset,
           _hip.MelfResult.pos.offset, _hip.MelfResult.value.offset, C.sizeof(_hip.MelfMatchInfo), _hip.MelfMatchInfo.tiles.offset]
    assert got == exp
    assert _hip.RESULT_DTYPE.itemsize == C.sizeof(_hip.MelfResult)


@pytest.mark.parametrize('sd', ['sample-images1', 'sample-images2'])
def test_params_loader_and_masks_match_oracle(sd):
    pfile = os.path.join(GOLDEN, sd, 'params.yml')
    p = _params.load(pfile)
    op = po.Params(pfile)
    assert p.dial_names == op.names
    assert p.dials_template_size == op.template_size
    assert (p.meter_rect.top_left, p.meter_rect.bottom_right) == op.meter_rect
    cp = p.to_c()
    ocp = op.c_params()
    assert (cp.th, cp.tw, cp.hue_shift, cp.ndials, cp.match_threshold) == (ocp.th, ocp.tw, ocp.hue_shift, ocp.ndials, ocp.match_threshold)
    assert list(cp.name_order)[:4] == list(op.name_order())
    for i in range(cp.ndials):
        (a, b) = (cp.dial[i], ocp.dial[i])
        assert (a.cx, a.cy, a.range_h, a.range_l, a.range_s, a.negative_momentum, a.angle_of_zero) == \
               (b.cx, b.cy, b.range_h, b.range_l, b.range_s, b.negative_momentum, b.angle_of_zero)
    assert list(cp.needle_lo) == [116, 35, 95] and list(cp.needle_hi) == [134, 125, 165]
    assert np.array_equal(_hip.build_dial_masks(cp), op.masks())
    # blob round trip
    blob = _engine.make_blob(p)
    back = _hip.blob_params(blob)
    assert bytes(back) == bytes(cp)
    assert np.array_equal(_engine.load_template(p), op.load_template())


def test_dial_masks_random_geometries_match_oracle():
    rng = np.random.default_rng(3)
    for _ in range(60):
        cp = _hip.MelfParams()
        cp.abi_version = _hip.ABI_VERSION
        (cp.th, cp.tw, cp.ndials) = (int(rng.integers(60, 130)), int(rng.integers(60, 200)), 3)
        cent, diam, dist, thick = [], [], [], []
        for d in range(3):
            dl = cp.dial[d]
            dl.diameter = int(rng.integers(4, 20))
            dl.dist_from_center = int(rng.integers(0, 6))
            dl.circle_thickness = int(rng.integers(1, 11))
            r_out = int(np.rint(dl.diameter / 2.0)) + dl.dist_from_center + dl.circle_thickness
            dl.cx = float(np.round(rng.uniform(r_out + 2, cp.tw - r_out - 3), 1))
            dl.cy = float(np.round(rng.uniform(r_out + 2, cp.th - r_out - 3), 1))
            if rng.random() < 0.3:
                (dl.cx, dl.cy) = (float(int(dl.cx)) + 0.5, float(int(dl.cy)) + 0.5)  # half-to-even rounding
            cent.append((dl.cx, dl.cy)); diam.append(dl.diameter); dist.append(dl.dist_from_center)
            thick.append(dl.circle_thickness)
        got = _hip.build_dial_masks(cp)
        exp = np.zeros_like(got)
        po.lib().orc_build_dial_masks(cp.th, cp.tw, 3, po._ptr(np.array(cent, np.float64)),
                                      po._ptr(np.array(diam, np.int32)), po._ptr(np.array(dist, np.int32)),
                                      po._ptr(np.array(thick, np.int32)), po._ptr(exp))
        assert np.array_equal(got, exp)


def test_load_errors(tmp_path):
    with pytest.raises(_params.LoadError, match='Cannot load YAML data from'):
        _params.load(str(tmp_path / 'missing.yml'))
    bad = tmp_path / 'list.yml'
    bad.write_text('- 1\n- 2\n')
    with pytest.raises(_params.LoadError, match='Not a valid parameters file'):
        _params.load(str(bad))
    good = open(os.path.join(GOLDEN, 'sample-images1', 'params.yml')).read()
    nofile = tmp_path / 'p.yml'
    nofile.write_text(good)
    with pytest.raises(_params.LoadError, match='File not found'):
        _params.load(str(nofile))
    wrong = tmp_path / 'w.yml'
    wrong.write_text(good.replace('hue_shift: 128', 'hue_shift: "128"'))
    (tmp_path / 'dials_gray.png').write_bytes(open(os.path.join(GOLDEN, 'sample-images1', 'dials_gray.png'), 'rb').read())
    with pytest.raises(_params.LoadError, match='hue_shift is not int'):
        _params.load(str(wrong))
    wrong.write_text(good.replace('center: [37.3, 63.4]', 'center: [37, 63.4]'))
    with pytest.raises(_params.LoadError, match='Item 0 in center is not float'):
        _params.load(str(wrong))


def test_hls_color_range_vectors():
    import json
    with open(os.path.join(GOLDEN, 'pure_fn_vectors.json')) as fp:
        vec = json.load(fp)
    for (c, g, lo, hi) in vec['get_range']:
        assert HlsColor(*c).get_range(HlsColor(*g)) == (HlsColor(*lo), HlsColor(*hi))


def test_exception_messages():
    e = exceptions.DialsNotFoundError('f.jpg', extra_info={'match val': 17495704.0})
    assert e.get_message() == 'Dials not found (match val = 17495704.0)'
    assert str(e) == 'Dials not found from file: f.jpg (match val = 17495704.0)'
    assert exceptions.NeedleContoursNotFoundError(extra_info={'dial': '0.01'}).get_message() == \
        'Cannot find needle contours of a dial (dial = 0.01)'
    assert exceptions.ImageLoadingError('x').get_message() == 'Unable to load image'
    assert isinstance(e, ValueError) and isinstance(exceptions.ImageLoadingError(), IOError)


def test_result_to_python_all_statuses():
    names = ['0.0001', '0.001', '0.01', '0.1']
    r = np.zeros(1, _hip.RESULT_DTYPE)[0]
    r['pos'][:4] = [6.2306, 3.3, 5.1, 2.4]
    r['value'] = 253.62306
    (vals, err) = _engine.result_to_python(r, names, 'f')
    assert err is None and vals == {'0.0001': 6.2306, '0.001': 3.3, '0.01': 5.1, '0.1': 2.4, 'value': 253.62306}
    r['status'] = _hip.FRAME_DIALS_NOT_FOUND
    r['match_val'] = 0.0
    assert _engine.result_to_python(r, names, 'f')[1].get_message() == 'Dials not found (match val = 0.0)'
    r['status'] = _hip.FRAME_NEEDLE_CONTOURS_NOT_FOUND
    r['failed_dial'] = 2
    assert _engine.result_to_python(r, names, 'f')[1].get_message() == 'Cannot find needle contours of a dial (dial = 0.01)'
    r['status'] = _hip.FRAME_ANGLE_UNDETERMINED
    r['unreadable_mask'] = 0b1010
    assert _engine.result_to_python(r, names, 'f')[1].get_message() == \
        'Cannot determine angle of a dial (unreadable dials = 0.001, 0.1)'


def test_result_to_python_debug_dial_positions(monkeypatch):
    """DEBUG only: 'dial positions' (the dials that were read, sorted by name, '{:.2f}') comes before
    'unreadable dials' (reference: meterelf/_reading.py:98-106)."""
    from meterelf_amd import _debug
    monkeypatch.setattr(_debug, 'DEBUG', {'masks'})
    names = ['0.0001', '0.001', '0.01', '0.1']
    r = np.zeros(1, _hip.RESULT_DTYPE)[0]
    r['pos'][:4] = [6.2306, 3.3, 5.105, 2.4]
    r['status'] = _hip.FRAME_ANGLE_UNDETERMINED
    r['unreadable_mask'] = 0b1010
    err = _engine.result_to_python(r, names, 'f')[1]
    assert list(err.extra_info) == ['dial positions', 'unreadable dials']
    assert err.get_message() == ('Cannot determine angle of a dial (dial positions =  (0.0001: 6.23 | 0.01: 5.11), '
                                 'unreadable dials = 0.001, 0.1)')


def test_records_to_python_equals_per_record_conversion():
    """The whole-array conversion of the file API gives what result_to_python gives record by record."""
    names = ['0.0001', '0.001', '0.01', '0.1']
    rng = np.random.default_rng(5)
    recs = np.zeros(40, _hip.RESULT_DTYPE)
    recs['status'] = rng.integers(0, 4, 40)
    recs['pos'] = rng.uniform(0, 10, (40, _hip.MAX_DIALS))
    recs['value'] = rng.uniform(0, 1000, 40)
    recs['match_val'] = rng.uniform(0, 2e7, 40).astype(np.float32)
    recs['failed_dial'] = rng.integers(0, 4, 40)
    recs['unreadable_mask'] = rng.integers(1, 16, 40)
    ok = (rng.integers(0, 5, 40) > 0).tolist()
    files = ['f%d' % i for i in range(40)]
    got = _engine.records_to_python(recs, ok, names, files)
    assert {_hip.FRAME_OK, _hip.FRAME_DIALS_NOT_FOUND, _hip.FRAME_NEEDLE_CONTOURS_NOT_FOUND,
            _hip.FRAME_ANGLE_UNDETERMINED} == set(recs['status'].tolist())
    for i in range(40):
        if not ok[i]:
            assert got[i] is None
            continue
        (vals, err) = _engine.result_to_python(recs[i], names, files[i])
        assert got[i][0] == vals and list(got[i][0]) == list(vals)
        assert all(type(v) is float for v in got[i][0].values())
        assert (got[i][1] is None) == (err is None)
        if err is not None:
            assert type(got[i][1]) is type(err) and got[i][1].get_message() == err.get_message()
            assert got[i][1].filename == err.filename


def test_records_to_items_equals_per_record_conversion():
    """The fused conversion of an all-GPU chunk (records -> MeterImageData in one comprehension) gives what the
    record-by-record path gives: same values, key order, float types, error classes and messages."""
    from meterelf_amd import MeterImageData
    names = ['0.0001', '0.001', '0.01', '0.1']
    rng = np.random.default_rng(6)
    n = 200
    recs = np.zeros(n, _hip.RESULT_DTYPE)
    recs['status'] = rng.choice([0, 0, 0, 0, 1, 2, 3], n)
    recs['pos'] = rng.uniform(0, 10, (n, _hip.MAX_DIALS))
    recs['value'] = rng.uniform(0, 1000, n)
    recs['match_val'] = rng.uniform(0, 2e7, n).astype(np.float32)
    recs['failed_dial'] = rng.integers(0, 4, n)
    recs['unreadable_mask'] = rng.integers(1, 16, n)
    files = ['f%d' % i for i in range(n)]
    for ok in ([True] * n, (rng.integers(0, 6, n) > 0).tolist()):
        got = _engine.records_to_items(recs, ok, names, files, MeterImageData)
        for i in range(n):
            if not ok[i]:
                assert got[i] is None
                continue
            (vals, err) = _engine.result_to_python(recs[i], names, files[i])
            item = got[i]
            assert item.filename == files[i] and item.meter_values == vals and list(item.meter_values) == list(vals)
            assert item.value == vals.get('value') and all(type(v) is float for v in item.meter_values.values())
            assert (item.error is None) == (err is None)
            if err is not None:
                assert type(item.error) is type(err) and item.error.get_message() == err.get_message()
    three = _engine.records_to_items(recs[:5], [True] * 5, names[:3], files[:5], MeterImageData)  # not four dials: no 'value'
    assert all('value' not in it.meter_values for it in three if it.error is None)


class _TwoInFlightReader:
    """A CPU stand-in with the file-name surface of MeterReader: up to two lists in flight, drained before anything else
    uses it (the contract get_meter_values relies on); readings come from the file NAME."""
    log = []

    def __init__(self, params, device=0, blob=None):
        self.dial_names = params.dial_names
        self.flight = []
        self.collected = []
        self.closed = False

    def close(self):
        assert not self.flight, 'closed with a list in flight'
        self.closed = True
        _TwoInFlightReader.log.append('close')

    def _records(self, paths):
        r = np.zeros(len(paths), _hip.RESULT_DTYPE)
        ok = np.ones(len(paths), bool)
        for (i, p) in enumerate(paths):
            k = int(os.path.basename(p).split('.')[0][1:])
            if k % 50 == 17:
                ok[i] = False          # a file for the host branch
            elif k % 7 == 3:
                r[i]['status'] = _hip.FRAME_DIALS_NOT_FOUND
                r[i]['match_val'] = k
            else:
                r[i]['pos'][:4] = [k % 10, (k // 10) % 10, (k // 100) % 10, (k // 1000) % 10]
                r[i]['value'] = float(k)
        return (r, ok)

    def read_jpeg_paths_batch(self, paths):
        assert not self.flight, 'one-piece call with a list in flight'
        return self._records(paths)

    def read_jpeg_paths_begin(self, paths):
        assert len(self.flight) < _hip.FILES_IN_FLIGHT_MAX, 'one list too many in flight'
        self.flight.append(list(paths))
        _TwoInFlightReader.log.append('begin %d' % len(paths))

    def read_jpeg_paths_end(self):
        paths = self.collected.pop(0) if self.collected else self.flight.pop(0)
        _TwoInFlightReader.log.append('end')
        return self._records(paths)

    def jpeg_paths_in_flight(self):
        return len(self.flight) + len(self.collected)

    def drain_jpeg_paths(self):
        self.collected += self.flight
        self.flight = []
        _TwoInFlightReader.log.append('drain')

    def discard_jpeg_paths(self):
        self.flight = []
        self.collected = []
        _TwoInFlightReader.log.append('discard')

    def read_many(self, images, cropped=None):
        assert not self.flight, 'host-decoded frames with a list in flight'
        r = np.zeros(len(images), _hip.RESULT_DTYPE)
        for (i, img) in enumerate(images):
            r[i]['pos'][:4] = [1, 2, 3, 4]
            r[i]['value'] = float(img[0, 0, 0]) + 0.5
        return list(r)


def test_get_meter_values_keeps_two_chunks_in_flight(tmp_path, monkeypatch):
    """The loop of get_meter_values with a reader that takes two lists in flight: results in input order, the next chunks
    begun before a chunk's records are touched, the reader drained before host-decoded frames go through it, nothing left
    in flight when the consumer stops early or the list ends."""
    from PIL import Image

    from meterelf_amd import _api
    monkeypatch.setattr(_api, 'MeterReader', _TwoInFlightReader)
    monkeypatch.setenv('METERELF_BATCH', '100')
    monkeypatch.setenv('METERELF_BATCH_FIRST', '0')     # every chunk full-sized (the ramp has its own test below)
    pfile = os.path.join(GOLDEN, 'sample-images2', 'params.yml')
    files = []
    for k in range(1, 731):
        f = str(tmp_path / ('f%d.png' % k))
        if k % 50 == 17:  # the host branch really decodes these
            Image.fromarray(np.full((4, 4, 3), k % 250, np.uint8)).save(f)
        files.append(f)
    _TwoInFlightReader.log = []
    got = list(_api.get_meter_values(pfile, files))
    assert [r.filename for r in got] == files
    for (k, r) in zip(range(1, 731), got):
        if k % 50 == 17:
            assert r.error is None and r.value == (k % 250) + 0.5      # through read_many (the PNG's red channel + 0.5)
        elif k % 7 == 3:
            assert r.value is None and 'Dials not found' in r.error.get_message()
        else:
            assert r.error is None and r.value == float(k) and r.meter_values[list(r.meter_values)[0]] == float(k % 10)
    log = _TwoInFlightReader.log
    assert log.count('close') == 1 and log[-1] == 'close'
    assert log.count('begin 100') == 6 and log.count('begin 30') == 1      # chunks 2..8 went through begin / end
    assert log.count('drain') >= 6                                          # every chunk with a host-branch file
    # as many lists as the library takes are begun right after the first chunk came back, before its records are used
    assert log[:3] == ['begin 100'] * _hip.FILES_IN_FLIGHT_MAX
    # a consumer that stops early leaves nothing in flight
    _TwoInFlightReader.log = []
    gen = _api.get_meter_values(pfile, files)
    first = [next(gen) for _ in range(150)]
    gen.close()
    assert [r.filename for r in first] == files[:150]
    assert 'discard' in _TwoInFlightReader.log and _TwoInFlightReader.log[-1] == 'close'


def test_get_meter_values_first_results_after_32_names(tmp_path, monkeypatch):
    """The reference pulls one name and yields one result (meterelf/_api.py:22-33).  Here the chunk sizes ramp up 32, 64, 128 ...
    METERELF_BATCH and nothing is pulled from `filenames` while a short chunk's results are outstanding: an iterator that stalls
    after 40 names has had its first 32 results by then; a long list ends up in full-sized, pipelined chunks."""
    from meterelf_amd import _api
    monkeypatch.setattr(_api, 'MeterReader', _TwoInFlightReader)
    monkeypatch.delenv('METERELF_BATCH', raising=False)
    monkeypatch.delenv('METERELF_BATCH_FIRST', raising=False)
    pfile = os.path.join(GOLDEN, 'sample-images2', 'params.yml')
    pulled = []

    class Stalled(Exception):
        pass

    def names():
        for k in range(1, 41):
            if k % 50 == 17:
                continue
            pulled.append(k)
            yield str(tmp_path / ('f%d.jpg' % k))
        raise Stalled()     # stands for "blocks here": nothing after the 40th name may be needed for the first 32 results

    _TwoInFlightReader.log = []
    gen = _api.get_meter_values(pfile, names())
    first = [next(gen) for _ in range(32)]
    assert len(pulled) == 32 and [os.path.basename(r.filename) for r in first] == ['f%d.jpg' % k for k in pulled]
    with pytest.raises(Stalled):
        next(gen)
    # sizes: 32, 64, ... 1024, then full chunks through begin / end with the library's look-ahead
    files = [str(tmp_path / ('f%d.jpg' % k)) for k in range(1, 6001) if k % 50 != 17]
    sizes = [len(c) for c in _api._chunks(files, 1024, 32)]
    assert sizes[:6] == [32, 64, 128, 256, 512, 1024] and set(sizes[6:-1]) == {1024} and sum(sizes) == len(files)
    assert [len(c) for c in _api._chunks(files[:100], 1024, 0)] == [100] and [len(c) for c in _api._chunks(files[:70], 16, 32)] == [16, 16, 16, 16, 6]
    _TwoInFlightReader.log = []
    got = list(_api.get_meter_values(pfile, files))
    assert [r.filename for r in got] == files
    log = _TwoInFlightReader.log
    assert log.count('begin 1024') >= 3 and not any(x.startswith('begin') and x != 'begin 1024' and not x.startswith('begin %d' % sizes[-1]) for x in log)


class _PerDeviceReader(_TwoInFlightReader):
    """The same stand-in, remembering which device it was made for and what it was given."""
    made = []

    def __init__(self, params, device=0, blob=None):
        super().__init__(params, device, blob)
        self.device = device
        self.seen = []
        _PerDeviceReader.made.append(self)

    def read_jpeg_paths_batch(self, paths):
        self.seen.append(list(paths))
        return super().read_jpeg_paths_batch(paths)

    def read_jpeg_paths_begin(self, paths):
        self.seen.append(list(paths))
        return super().read_jpeg_paths_begin(paths)


@pytest.mark.parametrize('devices', ['0,1', '0,1,2', '0,1,2,3,4,5,6,7', '0,0'])
def test_get_meter_values_over_several_devices(tmp_path, monkeypatch, devices):
    """The reference API over all GPUs of a node in one process (meterelf/_api.py:16-33: one list in, results in input order):
    chunks dealt round-robin over METERELF_DEVICES, one reader and one thread per entry; results in input order; a host-branch
    file in the middle of a chunk; a consumer that stops early leaves no worker running and nothing in flight; a worker's
    exception surfaces in the consumer."""
    import threading

    from PIL import Image

    from meterelf_amd import _api
    monkeypatch.setattr(_api, 'MeterReader', _PerDeviceReader)
    monkeypatch.setenv('METERELF_BATCH', '50')
    monkeypatch.setenv('METERELF_BATCH_FIRST', '0')   # full-sized chunks from the start (the ramp has its own test)
    monkeypatch.setenv('METERELF_DEVICES', devices)
    dev = [int(x) for x in devices.split(',')]
    pfile = os.path.join(GOLDEN, 'sample-images2', 'params.yml')
    files = []
    for k in range(1, 1238):
        f = str(tmp_path / ('f%d.png' % k))
        if k % 50 == 17:
            Image.fromarray(np.full((4, 4, 3), k % 250, np.uint8)).save(f)
        files.append(f)
    _PerDeviceReader.made = []
    def workers():
        return [t.name for t in threading.enumerate() if t.name.startswith('meterelf-dev')]
    got = list(_api.get_meter_values(pfile, files))
    assert not workers(), 'device workers still running'
    assert [r.filename for r in got] == files
    for (k, r) in zip(range(1, 1238), got):
        if k % 50 == 17:
            assert r.error is None and r.value == (k % 250) + 0.5
        elif k % 7 == 3:
            assert r.value is None and 'Dials not found' in r.error.get_message()
        else:
            assert r.error is None and r.value == float(k)
    readers = _PerDeviceReader.made
    assert [r.device for r in sorted(readers, key=lambda r: files.index(r.seen[0][0]))] == dev   # one reader per entry, on its device
    assert all(r.closed and not r.flight and not r.collected for r in readers)
    chunks = [files[i:i + 50] for i in range(0, len(files), 50)]
    for (w, r) in enumerate(sorted(readers, key=lambda r: files.index(r.seen[0][0]))):
        assert r.seen == chunks[w::len(dev)], w            # worker w got chunks w, w + D, ... in order
    # a consumer that stops early
    _PerDeviceReader.made = []
    gen = _api.get_meter_values(pfile, files)
    first = [next(gen) for _ in range(120)]
    gen.close()
    assert [r.filename for r in first] == files[:120]
    assert not workers()
    assert all(r.closed and not r.flight and not r.collected for r in _PerDeviceReader.made)
    # a worker that fails: the exception reaches the consumer at that chunk's position, the other workers are collected
    class _Boom(_PerDeviceReader):
        def _records(self, paths):
            if any(os.path.basename(p) == 'f333.png' for p in paths):
                raise RuntimeError('device lost')
            return super()._records(paths)
    monkeypatch.setattr(_api, 'MeterReader', _Boom)
    _PerDeviceReader.made = []
    seen = []
    with pytest.raises(RuntimeError, match='device lost'):
        for r in _api.get_meter_values(pfile, files):
            seen.append(r.filename)
    assert seen == files[:300]          # everything before the failing chunk (files 301..350) was delivered, in order
    assert not workers()


def test_context_cache_key_and_fork_hook(monkeypatch):
    """The cache of idle GPU contexts: the key carries the device and the environment switches the library reads at context
    creation (a changed MELF_MATCH must not get a stale context back); a forked child forgets the parent's contexts without
    calling into the library."""
    from meterelf_amd import _api
    p = _params.load(os.path.join(GOLDEN, 'sample-images1', 'params.yml'))
    blob = _engine.make_blob(p)
    k0 = _api._reader_key(p, blob, 0)
    assert k0 == _api._reader_key(p, blob, 0) and k0 != _api._reader_key(p, blob, 1)
    monkeypatch.setenv('MELF_MATCH', 'dot4')
    assert _api._reader_key(p, blob, 0) != k0
    monkeypatch.delenv('MELF_MATCH')
    assert _api._reader_key(p, blob, 0) == k0

    class _Ctx:
        _h = 123

    class _R:
        ctx = _Ctx()
        _crop_ctx = {(1, 1): _Ctx()}
    r = _R()
    monkeypatch.setitem(_api._idle_readers, b'x', r)
    _api._forget_contexts_in_child()
    assert not _api._idle_readers and r.ctx._h is None and r._crop_ctx[(1, 1)]._h is None
    # unset: ONE device (LOCAL_RANK's under torchrun, else 0) -- a rank of a process-per-GPU job never opens its neighbours' GPUs
    monkeypatch.delenv('METERELF_DEVICES', raising=False)
    monkeypatch.delenv('LOCAL_RANK', raising=False)
    assert _api._device_list() == [0]
    monkeypatch.setenv('LOCAL_RANK', '5')
    assert _api._device_list() == [5 % max(1, _hip.device_count())]
    monkeypatch.setenv('METERELF_DEVICES', 'all')
    assert _api._device_list() == list(range(max(1, _hip.device_count())))
    monkeypatch.setenv('METERELF_DEVICES', '2, 0,0')
    assert _api._device_list() == [2, 0, 0]
    # the switch-interval override is counted: overlapping fan-outs share it, the last one out restores
    import sys
    before = sys.getswitchinterval()
    monkeypatch.setenv('METERELF_SWITCH_INTERVAL', '0.0004')
    assert _api._switch_interval_enter() and _api._switch_interval_enter()
    assert abs(sys.getswitchinterval() - 0.0004) < 1e-9
    _api._switch_interval_exit()
    assert abs(sys.getswitchinterval() - 0.0004) < 1e-9      # one generator still running
    _api._switch_interval_exit()
    assert sys.getswitchinterval() == before
    monkeypatch.setenv('METERELF_SWITCH_INTERVAL', '0')
    assert not _api._switch_interval_enter() and sys.getswitchinterval() == before


def test_no_gpu_means_loud_failure():
    if _hip.device_count() > 0:
        pytest.skip('a GPU is visible')
    p = _params.load(os.path.join(GOLDEN, 'sample-images1', 'params.yml'))
    with pytest.raises(_hip.HipError, match='no CPU fallback'):
        _hip.Context(_engine.make_blob(p))


def test_product_never_imports_the_oracle():
    for f in glob.glob(os.path.join(ROOT, 'meterelf_amd', '**', '*'), recursive=True):
        if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
            assert 'oracle' not in open(f).read().lower().replace('the cpu oracle', ''), f


@pytest.mark.parametrize('sd,count', [('sample-images1', 81), ('sample-images2', 223)])
def test_cli_host_logic_against_golden_with_oracle_backend(sd, count, capsys, monkeypatch):
    """API + CLI host logic (chunking, decode, error conversion, formatting) on CPU:
    the GPU context is replaced -- in this test only -- by the oracle."""
    monkeypatch.setattr(_api, 'MeterReader', helpers.OracleReader)
    monkeypatch.setenv('METERELF_BATCH', '7')
    files = sorted(glob.glob(os.path.join(GOLDEN, sd, '*.jpg')))
    cwd = os.getcwd()
    os.chdir(os.path.join(GOLDEN, sd))
    try:
        _main.main(['meterelf', 'params.yml'] + [os.path.basename(f) for f in files] + ['nonexistent.jpg'])
    finally:
        os.chdir(cwd)
    out = capsys.readouterr()
    assert out.err == ''
    lines = out.out.splitlines()
    with open(os.path.join(GOLDEN, sd + '_stdout.txt')) as fp:
        expected = fp.read().splitlines()
    assert len(lines) == count + 1
    assert lines[-1] == 'nonexistent.jpg: UNKNOWN Unable to load image'
    diff = [(g, e) for (g, e) in zip(lines, expected) if g != e]
    assert all('20180814021310-00-e02.jpg' in g for (g, _e) in diff) and len(diff) <= 1


def test_cli_usage_message():
    with pytest.raises(SystemExit) as e:
        _main.main(['meterelf'])
    assert str(e.value) == 'Usage: meterelf PARAMETERS_FILE [IMAGE_FILE...]'


def test_debug_mode_reraises(monkeypatch):
    from meterelf_amd import _debug
    monkeypatch.setattr(_api, 'MeterReader', helpers.OracleReader)
    monkeypatch.setattr(_debug, 'DEBUG', {'1'})
    pfile = os.path.join(GOLDEN, 'sample-images1', 'params.yml')
    f = os.path.join(GOLDEN, 'sample-images1', '20180814021309-01-e01.jpg')
    with pytest.raises(exceptions.DialsNotFoundError) as e:
        list(_api.get_meter_values(pfile, [f]))
    assert e.value.get_message() == 'Dials not found (match val = 0.0)'  # reference tests/test_meterelf.py:164-167


def test_debug_mode_stdout_suffix(monkeypatch, capsys):
    """reference tests/test_meterelf.py:170-188: in DEBUG mode the line carries repr(meter_values)."""
    import json
    from meterelf_amd import _debug
    monkeypatch.setattr(_api, 'MeterReader', helpers.OracleReader)
    monkeypatch.setattr(_debug, 'DEBUG', {'1'})
    f = os.path.join(GOLDEN, 'sample-images1', '20180814215230-01-e136.jpg')
    _main.main(['meterelf', os.path.join(GOLDEN, 'sample-images1', 'params.yml'), f])
    out = capsys.readouterr()
    basic = f + ': 253.623'
    assert out.out.startswith(basic) and out.err == ''
    data = json.loads(out.out[len(basic):].replace("'", '"').strip())
    assert set(data) == {'0.0001', '0.001', '0.01', '0.1', 'value'}
    assert abs(data['0.0001'] - 6.23) < 0.005 and abs(data['0.001'] - 3.3) < 0.05
    assert abs(data['0.01'] - 5.1) < 0.05 and abs(data['0.1'] - 2.4) < 0.05
    assert abs(data['value'] - 253.62306) < 0.000005


def test_colour_template_grey_conversion_follows_the_file_format(tmp_path):
    """cv2.imread(..., IMREAD_GRAYSCALE) of a COLOUR template: libpng's 15-bit weights for PNG (palettes expanded first),
    cvtColor's 14-bit weights for formats decoded to BGR; grey files are taken as they are."""
    import shutil
    import yaml
    from PIL import Image
    src = os.path.join(GOLDEN, 'sample-images1')
    with open(os.path.join(src, 'params.yml')) as fp:
        data = yaml.safe_load(fp)
    (tw, th) = data['dials_template_size']
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, size=(th, tw, 3), dtype=np.uint8)
    (r, g, b) = (rgb[..., 0].astype(np.int64), rgb[..., 1].astype(np.int64), rgb[..., 2].astype(np.int64))

    def load(name, save):
        d = tmp_path / name
        d.mkdir()
        data2 = dict(data)
        data2['dials_template'] = name + '.img'
        save(str(d / data2['dials_template']))
        with open(d / 'params.yml', 'w') as fp:
            yaml.safe_dump(data2, fp)
        return _engine.load_template(_params.load(str(d / 'params.yml')))

    from tests.helpers import libpng_rgb_to_gray
    png = load('png', lambda f: Image.fromarray(rgb, 'RGB').save(f, 'PNG'))
    assert np.array_equal(png, ((r * 9797 + g * 19234 + b * 3737) >> 15).astype(np.uint8))
    # ... which is what libpng itself produces for the call OpenCV 3.4 makes (png_set_rgb_to_gray(png, 1, 0.299, 0.587))
    ref = libpng_rgb_to_gray(str(tmp_path / 'png' / 'png.img'))
    if ref is not None:
        assert np.array_equal(png, ref)
    bmp = load('bmp', lambda f: Image.fromarray(rgb, 'RGB').save(f, 'BMP'))
    assert np.array_equal(bmp, ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))
    assert (png != bmp).any()      # the two formulas are not the same function
    pal = Image.fromarray(rgb, 'RGB').quantize(64)
    prgb = np.asarray(pal.convert('RGB'), dtype=np.int64)
    ppng = load('pal', lambda f: pal.save(f, 'PNG'))
    assert np.array_equal(ppng, ((prgb[..., 0] * 9797 + prgb[..., 1] * 19234 + prgb[..., 2] * 3737) >> 15).astype(np.uint8))
    ref = libpng_rgb_to_gray(str(tmp_path / 'pal' / 'pal.img'))
    if ref is not None:
        assert np.array_equal(ppng, ref)
    grey = rng.integers(0, 256, size=(th, tw), dtype=np.uint8)
    assert np.array_equal(load('grey', lambda f: Image.fromarray(grey, 'L').save(f, 'PNG')), grey)


def test_no_scratch_in_the_hot_path_kernels():
    """Code-object metadata of the built library (tools/kernel_meta.py reads the amdhsa notes; no GPU): no kernel that
    default dispatch launches on the reading path, the fused-mask stage or the JPEG stage has a private segment
    (private_segment_fixed_size == 0: no spilled vector registers in scratch, no dynamically indexed local array) -- a kernel
    that spills still computes the right thing, so only a check of the build notices (round 3's k_dials carried 168 bytes per
    lane, k_match_gen<8> 592)."""
    import re
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import kernel_meta
    meta = kernel_meta.kernel_metadata()
    assert len(meta) > 40, 'no kernel metadata found in libmeterelf_hip.so'
    hot = []
    for (name, d) in meta.items():
        fused = re.search(r'k_fused_mask_lutILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name)
        if fused:
            # the launch shapes launch_lut_v picks by default: interval tables <6..8, 1024, 1, 8>, bit tables / generic <0..3, 1024, 1, 4>,
            # ties <4, 512, 0, 4> (the other instantiations are MELF_FUSED_CONFIG experiments)
            (v, t, pf, wps) = (int(x) for x in fused.groups())
            if not ((v >= 6 and (t, pf, wps) == (1024, 1, 8)) or (v <= 3 and (t, pf, wps) == (1024, 1, 4)) or (v == 4 and (t, pf, wps) == (512, 0, 4))):
                continue
        elif not any(k in name for k in ('k_prep_lplane', 'k_match_mfma', 'k_match_gen', 'k_dials', 'k_jpeg_huff', 'k_jpeg_idct',
                                         'k_jpeg_color', 'k_bgr2hls', 'k_fused_mask')):
            continue
        hot.append(name)
        assert d.get('private_segment_fixed_size', 0) == 0, (name, d)
        assert d.get('wavefront_size', 64) == 64
    kinds = {k: sum(1 for n in hot if k in n) for k in ('k_prep_lplane', 'k_match_mfma', 'k_match_gen', 'k_dials', 'k_fused_mask_lut', 'k_jpeg_huff')}
    assert kinds['k_prep_lplane'] == 2 and kinds['k_match_mfma'] == 12 and kinds['k_match_gen'] == 7 and kinds['k_dials'] == 12, kinds
    assert kinds['k_fused_mask_lut'] >= 8 and kinds['k_jpeg_huff'] >= 4, kinds
    assert not any('k_colsum' in n for n in meta)       # the window sums are added up by the match waves since round 4


def test_import_only_and_run_as_script():
    """The reference's tests/test_main.py:9-22: importing the package's __main__ module does not run main(); running the
    package as a script (`python -m meterelf_amd`, the reference's integration-tests/test_all_sample_images:20) calls it."""
    import runpy
    from unittest.mock import patch

    import meterelf_amd
    with patch.object(_main, 'main') as main_func_mock:
        from meterelf_amd import __main__ as main_mod
        main_func_mock.assert_not_called()
        assert main_mod.__name__ == '{}.__main__'.format(meterelf_amd.__name__)
        del sys.modules[main_mod.__name__]   # as the reference does: no "found in sys.modules" warning below
    with patch.object(_main, 'main') as main_func_mock:
        runpy.run_module(meterelf_amd.__name__, run_name='__main__')
        main_func_mock.assert_called_with()


def test_path_table_addresses_and_embedded_nul():
    """The file-name table the C entry points get: one NUL-terminated string per name; a name with an embedded NUL is refused
    the way open() refuses it (a C string would end there and silently name another file)."""
    import ctypes as C
    names = ['a.jpg', 'dir/b c.jpg', 'ü.jpg']
    (addr, keep) = _hip._path_table(names)
    assert [C.string_at(int(a)) for a in addr] == [os.fsencode(n) for n in names]
    (addr, keep) = _hip._path_table([b'x.jpg', 'y.jpg'])
    assert [C.string_at(int(a)) for a in addr] == [b'x.jpg', b'y.jpg']
    with pytest.raises(ValueError, match='embedded null byte'):
        _hip._path_table(['good.jpg\0x', 'other.jpg'])


class _TimedDeviceReader:
    """A CPU stand-in for one GPU's file-name pipeline: a call's records come back no earlier than its latency after its
    _begin and no earlier than one device period after the previous call's (a pipelined device: the rates measured on the
    GPU box, profiles/r05/api_host_breakdown.txt); the waiting sleeps, i.e. releases the interpreter lock like the
    library's _end does.  Records are prepared once: what the test times is the host logic, not the stand-in."""
    LATENCY = 0.016     # s, _begin -> records ready (read + enqueue + kernels of a 1024-file call)
    PERIOD = 0.008      # s per call the device sustains

    def __init__(self, params, device=0, blob=None):
        self.dial_names = params.dial_names
        self.device = device
        self.flight = []       # (ready time, n)
        self.last_ready = 0.0
        self._cache = {}

    def close(self):
        assert not self.flight

    def _records(self, n):
        if n not in self._cache:
            r = np.zeros(n, _hip.RESULT_DTYPE)
            k = np.arange(n)
            r['value'] = 100.0 + (k % 800) + 0.125
            for d in range(4):
                r['pos'][:, d] = (k // 10 ** d) % 10 + 0.25
            self._cache[n] = (r, np.ones(n, bool))
        return self._cache[n]

    def read_jpeg_paths_begin(self, paths):
        assert len(self.flight) < _hip.FILES_IN_FLIGHT_MAX
        now = time.perf_counter()
        ready = max(now + self.LATENCY, self.last_ready + self.PERIOD)
        self.last_ready = ready
        self.flight.append((ready, len(paths)))

    def read_jpeg_paths_end(self):
        (ready, n) = self.flight.pop(0)
        delay = ready - time.perf_counter()
        if delay > 0:
            time.sleep(delay)
        return self._records(n)

    def read_jpeg_paths_batch(self, paths):
        self.read_jpeg_paths_begin(paths)
        return self.read_jpeg_paths_end()

    def jpeg_paths_in_flight(self):
        return len(self.flight)

    def drain_jpeg_paths(self):
        pass

    def discard_jpeg_paths(self):
        self.flight = []


def test_fan_out_throughput_scales_with_devices(monkeypatch):
    """The reference API over N GPUs in ONE process (meterelf/_api.py:16-33 -> _api._fan_out): the pool hands out one GPU, so the
    only multi-device evidence for this path is the host logic driven by stand-in devices that take the time a real pipeline
    takes per 1024-file chunk.  Throughput must grow with the devices until the interpreter (one thread at a time turns a
    chunk's records into MeterImageData objects) is the limit: 1 -> 2 devices nearly doubles, 2 -> 4 grows again, 8 stays
    within a quarter of 4 (past the ceiling the workers only hand the lock round); one device reaches its device's rate (the fan-out and the pipeline add no bubbles)."""
    from meterelf_amd import _api
    monkeypatch.setattr(_api, 'MeterReader', _TimedDeviceReader)
    monkeypatch.setenv('METERELF_BATCH', '1024')
    monkeypatch.setenv('METERELF_BATCH_FIRST', '0')   # the stand-in device takes a full chunk's time for any chunk: no ramp here
    pfile = os.path.join(GOLDEN, 'sample-images1', 'params.yml')
    names = ['/nowhere/f%06d.jpg' % i for i in range(48 * 1024)]
    device_rate = 1024 / _TimedDeviceReader.PERIOD

    def measure():
        rate = {}
        for ndev in (1, 2, 4, 8):
            monkeypatch.setenv('METERELF_DEVICES', ','.join(str(d) for d in range(ndev)))
            best = 0.0
            for rep in range(2):
                t0 = time.perf_counter()
                n = sum(1 for r in _api.get_meter_values(pfile, names) if r.error is None)
                best = max(best, n / (time.perf_counter() - t0))
                assert n == len(names)
            rate[ndev] = best
        ok = (rate[1] >= 0.80 * device_rate          # the pipeline keeps one device busy
              and rate[2] >= 1.4 * rate[1]           # (measured 1.6-1.9; the margins are for a loaded test box)
              and rate[4] >= 1.2 * rate[2]           # (measured 1.65-2.0)
              and rate[8] >= 0.7 * rate[4])
        return ok, rate
    # a timing test on a shared CPU box: a noisy neighbour may spoil one measurement, not three
    for attempt in range(3):
        (ok, rate) = measure()
        if ok:
            break
    assert ok, rate              # (8 against 4: at the interpreter's ceiling the workers only hand the lock round)
    # order and content survive the fan-out at full speed
    got = list(_api.get_meter_values(pfile, names[:5000]))
    assert [r.filename for r in got] == names[:5000] and got[1234].value == 100.0 + (1234 - 1024) % 800 + 0.125


def test_ctx_create_bcast_argument_checks():
    """melf_ctx_create_bcast (SURVEY 8b's melf_ctx_bcast): one RCCL rank per GPU -- a device listed twice is refused before anything
    is touched; without a GPU the call fails loudly like melf_ctx_create (no CPU fallback), and every out[i] stays NULL."""
    p = _params.load(os.path.join(GOLDEN, 'sample-images1', 'params.yml'))
    blob = _engine.make_blob(p)
    with pytest.raises(_hip.HipError, match='listed twice'):
        _hip.Context.create_bcast(blob, [0, 0])
    if _hip.device_count() == 0:
        with pytest.raises(_hip.HipError):
            _hip.Context.create_bcast(blob, [0])
    # the fan-out only broadcasts to DISTINCT devices; "0,0" (two contexts on one GPU) and a switched-off broadcast create one by one
    assert _api._acquire_readers_bcast(p, [0, 0]) == {} and _api._acquire_readers_bcast(p, [0]) == {}


def test_fan_out_uses_broadcast_created_readers(monkeypatch):
    """_fan_out with distinct devices: the readers come from ONE melf_ctx_create_bcast call (stand-in here), each worker gets the
    reader of its device, every reader is handed back (cache) or closed; when the broadcast fails the workers fall back to
    creating their own contexts and the results are the same."""
    made = []

    class _Ctx:
        def __init__(self, device):
            self.device = device
            self.closed = False

        def close(self):
            self.closed = True

    class _Reader(_TimedDeviceReader):
        LATENCY = 0.0
        PERIOD = 0.0

        def __init__(self, params, device=0, blob=None, ctx=None):
            super().__init__(params, device, blob)
            self.ctx = ctx
            self.from_bcast = ctx is not None
            self.closed = False
            made.append(self)

        def close(self):
            super().close()
            self.closed = True

    calls = []

    def fake_bcast(blob, devices):
        calls.append(list(devices))
        return [_Ctx(d) for d in devices]
    monkeypatch.setattr(_api, 'MeterReader', _Reader)
    monkeypatch.setattr(_api, '_REAL_READER', _Reader)       # "the real reader": the broadcast and the cache are in play
    monkeypatch.setattr(_hip.Context, 'create_bcast', staticmethod(fake_bcast))
    monkeypatch.setenv('METERELF_CTX_CACHE', '0')
    monkeypatch.setenv('METERELF_BATCH', '64')
    monkeypatch.setenv('METERELF_DEVICES', '0,1,2')
    pfile = os.path.join(GOLDEN, 'sample-images1', 'params.yml')
    names = ['/nowhere/f%05d.jpg' % i for i in range(1000)]
    got = list(_api.get_meter_values(pfile, names))
    assert [r.filename for r in got] == names and calls == [[0, 1, 2]]
    assert sorted(r.device for r in made) == [0, 1, 2] and all(r.from_bcast and r.closed for r in made)
    # broadcast not available: same results, contexts created one by one
    del made[:]
    del calls[:]

    def failing(blob, devices):
        calls.append(list(devices))
        raise _hip.HipError('no RCCL')
    monkeypatch.setattr(_hip.Context, 'create_bcast', staticmethod(failing))
    got2 = list(_api.get_meter_values(pfile, names))
    assert [(r.filename, r.value) for r in got2] == [(r.filename, r.value) for r in got] and calls == [[0, 1, 2]]
    assert sorted(r.device for r in made) == [0, 1, 2] and not any(r.from_bcast for r in made) and all(r.closed for r in made)
