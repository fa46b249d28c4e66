"""SURVEY section 8 row f4 on the GPU: the DEBUG switch through the real HIP reader (no injected backend).

Mirrors the reference's own DEBUG tests (tests/test_meterelf.py:147-188) and pins the DEBUG-only
'dial positions' extra of DialAngleDeterminingError (meterelf/_reading.py:98-106) on a synthetic frame with
two unreadable dials; the expected text is built from the CPU oracle's record (checker only)."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture()
def debug_on(monkeypatch):
    from meterelf_amd import _debug, _hip
    if _hip.device_count() < 1:
        pytest.fail('GPU tests need an MI355X: no HIP device visible (no CPU fallback exists)')
    monkeypatch.setattr(_debug, 'DEBUG', {'1'})


@pytest.mark.parametrize('decode', ['gpu', 'host'])
def test_debug_reraises_dials_not_found(debug_on, monkeypatch, decode):
    """reference tests/test_meterelf.py:147-167: with DEBUG the per-image error propagates, message included."""
    from meterelf_amd import _api, exceptions
    monkeypatch.setenv('METERELF_DECODE', decode)
    pfile = os.path.join(GOLDEN, 'sample-images1', 'params.yml')
    f = os.path.join(GOLDEN, 'sample-images1', '20180814021309-01-e01.jpg')
    with pytest.raises(exceptions.DialsNotFoundError) as e:
        list(_api.get_meter_values(pfile, [f]))
    assert e.value.get_message() == 'Dials not found (match val = 0.0)'
    assert e.value.filename == f
    # the file after the failing one is never touched: a generator stops at the raise
    gen = _api.get_meter_values(pfile, [os.path.join(GOLDEN, 'sample-images1', '20180814215230-01-e136.jpg'), f, '/nonexistent.jpg'])
    first = next(gen)
    assert first.error is None and '{:07.3f}'.format(first.value) == '253.623'
    with pytest.raises(exceptions.DialsNotFoundError):
        next(gen)


def test_debug_stdout_suffix(debug_on, capsys):
    """reference tests/test_meterelf.py:170-188: the line carries repr(meter_values); e136 intermediate golden."""
    from meterelf_amd import _main
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


def _unreadable_frame():
    """A synthetic sample-images2 frame on which the oracle finds dials 0.01 and 0.1 unreadable."""
    from oracle import pyoracle as po
    from tests.test_gpu_parity import _good, synth_frames
    sd = 'sample-images2'
    pfile = os.path.join(GOLDEN, sd, 'params.yml')
    files = _good(sorted(glob.glob(os.path.join(GOLDEN, sd, '*.jpg'))))
    frames = synth_frames(files, 9, 3, shift=4, sigma=16.0)
    o = po.process_frames(frames[8:9], po.Params(pfile))[0]
    assert o.status == 3 and o.unreadable_mask == 0b1100
    return pfile, frames[8], o


def test_debug_unreadable_dials_carry_dial_positions(debug_on, tmp_path):
    from PIL import Image
    from meterelf_amd import _api, _params, exceptions
    from meterelf_amd._image import ImageFile
    from meterelf_amd._reading import get_meter_value
    (pfile, frame, o) = _unreadable_frame()
    names = ['0.0001', '0.001', '0.01', '0.1']
    # meterelf/_reading.py:98-106: readable dials sorted by name, '{:.2f}', then the unreadable list in dial order
    expected = ('Cannot determine angle of a dial (dial positions =  (0.0001: {:.2f} | 0.001: {:.2f}), '
                'unreadable dials = 0.01, 0.1)').format(o.pos[0], o.pos[1])
    png = str(tmp_path / 'noisy.png')   # lossless, so the file holds exactly the synthetic frame
    Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])).save(png)
    with pytest.raises(exceptions.DialAngleDeterminingError) as e:
        list(_api.get_meter_values(pfile, [png]))
    assert e.value.get_message() == expected and e.value.filename == png
    assert list(e.value.extra_info) == ['dial positions', 'unreadable dials']
    # the same through get_meter_value(imgf) (meterelf/_reading.py:19), which raises with or without DEBUG
    with pytest.raises(exceptions.DialAngleDeterminingError) as e2:
        get_meter_value(ImageFile(png, _params.load(pfile)))
    assert e2.value.get_message() == expected


def test_without_debug_no_dial_positions(tmp_path, monkeypatch):
    from PIL import Image
    from meterelf_amd import _api, _debug
    monkeypatch.setattr(_debug, 'DEBUG', set())
    (pfile, frame, o) = _unreadable_frame()
    png = str(tmp_path / 'noisy.png')
    Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])).save(png)
    (data,) = list(_api.get_meter_values(pfile, [png]))
    assert data.value is None and data.meter_values == {}
    assert data.error.get_message() == 'Cannot determine angle of a dial (unreadable dials = 0.01, 0.1)'
