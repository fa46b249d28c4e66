"""GPU parity tests proper: every HIP stage and the whole path, called through
the C ABI (meterelf_amd._hip -> libmeterelf_hip.so), against the CPU oracle on
the same inputs and against the reference's golden stdout.

Bars: bit-exact for bytes / masks / indices / float32 match values / digits;
dial positions and needle angles within 1e-9 (north star asks 1e-3) -- they are
float64 sums whose order differs between the wave reduction and Python's
left-to-right sum.
"""
import glob
import os
import re
import sys

import numpy as np
import pytest

from tests.helpers import hip_runtime

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POS_TOL = 1e-9
NOISY_MATCH_VAL = '20180814021310-00-e02.jpg'
REJECTED = ('20180814021309-01-e01.jpg', NOISY_MATCH_VAL)  # the two 'Dials not found' frames


def _good(files):
    return [f for f in files if os.path.basename(f) not in REJECTED]


@pytest.fixture(scope='module')
def env():
    from meterelf_amd import _hip
    if _hip.device_count() < 1:
        pytest.fail('GPU tests need an MI355X: no HIP device visible (no CPU fallback exists)')
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    out = {}
    for sd in ('sample-images1', 'sample-images2'):
        pfile = os.path.join(GOLDEN, sd, 'params.yml')
        out[sd] = dict(pfile=pfile, params=_params.load(pfile), oparams=po.Params(pfile),
                       files=sorted(glob.glob(os.path.join(GOLDEN, sd, '*.jpg'))))
        out[sd]['reader'] = MeterReader(out[sd]['params'])
    yield out
    for sd in ('sample-images1', 'sample-images2'):
        out[sd]['reader'].close()


def _compare_records(recs, ores, ndials=4, tag=''):
    for i in range(len(recs)):
        (r, o) = (recs[i], ores[i])
        assert int(r['status']) == o.status, (tag, i, int(r['status']), o.status)
        assert (int(r['match_x']), int(r['match_y'])) == (o.match_x, o.match_y), (tag, i)
        assert float(r['match_val']) == o.match_val, (tag, i)  # float32, bit-exact
        if o.status == 0:
            assert np.allclose(r['pos'][:ndials], list(o.pos)[:ndials], rtol=0, atol=POS_TOL), (tag, i)
            assert np.allclose(r['angle'][:ndials], list(o.angle)[:ndials], rtol=0, atol=POS_TOL), (tag, i)
            assert abs(float(r['value']) - o.value) < 1e-8, (tag, i)
            assert int(float(r['value'])) == int(o.value), (tag, i)  # the three dial digits
        elif o.status == 2:
            assert int(r['failed_dial']) == o.failed_dial, (tag, i)
        elif o.status == 3:
            assert int(r['unreadable_mask']) == o.unreadable_mask, (tag, i)


# ---------------------------------------------------------------- stages ----

@pytest.mark.parametrize('shape', [(250, 250), (135, 220), (7, 5), (3, 257), (64, 1030), (1, 1)])
def test_bgr2hls_bit_exact(env, shape):
    from oracle import pyoracle as po
    ctx = env['sample-images1']['reader'].ctx
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    bgr = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    # grey / saturated / near-grey pixels hit the diff <= eps and h wrap branches
    bgr[0, 0] = (7, 7, 7)
    if shape[1] > 2:
        bgr[0, 1] = (255, 0, 0)
        bgr[0, 2] = (10, 11, 10)
    got = ctx.bgr2hls(bgr)
    exp = po.bgr2hls(bgr, ctx.params.hue_shift)
    assert np.array_equal(got, exp)


def test_bgr2hls_exhaustive_sample(env):
    """Every (max, min) pair and many hue numerators: 2^21 structured triples."""
    from oracle import pyoracle as po
    ctx = env['sample-images1']['reader'].ctx
    v = np.arange(256, dtype=np.uint8)
    a = np.stack(np.meshgrid(v, v, v[::8], indexing='ij'), axis=-1).reshape(-1, 3)
    for perm in ([0, 1, 2], [2, 0, 1], [1, 2, 0]):
        bgr = np.ascontiguousarray(a[:, perm]).reshape(2048, -1, 3)
        assert np.array_equal(ctx.bgr2hls(bgr), po.bgr2hls(bgr, ctx.params.hue_shift))


def _blobby(rng, n, H, W):
    """Frames with needle-coloured blobs so that inRange + closing has work to do."""
    base = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
    small = rng.random((n, H // 8 + 1, W // 8 + 1)) < 0.35
    big = np.kron(small, np.ones((8, 8), bool))[:, :H, :W]
    red = np.array([40, 30, 200], np.uint8)  # BGR needle-ish red
    noise = rng.integers(-25, 25, size=(n, H, W, 3))
    blob = np.clip(red[None, None, None, :].astype(np.int64) + noise, 0, 255).astype(np.uint8)
    base[big] = blob[big]
    return base


@pytest.mark.parametrize('sd', ['sample-images1', 'sample-images2'])
@pytest.mark.parametrize('n,H,W', [(3, 640, 480), (2, 480, 640), (2, 37, 53), (1, 33, 100), (2, 70, 1920),
                                   (5, 3, 16), (3, 101, 48), (2, 67, 80), (1, 1, 32), (7, 40, 1040)])
def test_fused_mask_bit_exact(env, sd, n, H, W):
    from oracle import pyoracle as po
    e = env[sd]
    ctx = e['reader'].ctx
    p = ctx.params
    rng = np.random.default_rng(H * W + n)
    frames = _blobby(rng, n, H, W)
    got = ctx.hls_inrange_close(frames)
    lo, hi = list(p.needle_lo), list(p.needle_hi)
    nz = 0
    for f in range(n):
        exp = po.hls_inrange_close(frames[f], p.hue_shift, lo, hi)
        assert np.array_equal(got[f], exp), (f, np.argwhere(got[f] != exp)[:5])
        nz += int((exp > 0).sum())
    assert nz > 0 or H * W < 2000  # the test must exercise set pixels


def test_fused_mask_all_2_24_triples(env):
    """The table-driven fast path against the oracle's float path for EVERY BGR
    triple: each test pixel sits alone on a black background (spacing 4), so the
    3x3 closing returns exactly its own in-range bit."""
    from oracle import pyoracle as po
    ctx = env['sample-images1']['reader'].ctx
    p = ctx.params
    (lo, hi) = (list(p.needle_lo), list(p.needle_hi))
    side = 512
    per = side * side
    set_total = 0
    for chunk in range((1 << 24) // per):
        t = np.arange(chunk * per, (chunk + 1) * per, dtype=np.uint32)
        tri = np.stack([t & 255, (t >> 8) & 255, (t >> 16) & 255], axis=-1).astype(np.uint8).reshape(side, side, 3)
        img = np.zeros((side * 4, side * 4, 3), np.uint8)
        img[1::4, 1::4] = tri
        got = ctx.hls_inrange_close(img[None])[0]
        exp = po.hls_inrange_close(img, p.hue_shift, lo, hi)
        assert np.array_equal(got, exp), chunk
        set_total += int((exp[1::4, 1::4] > 0).sum())
        assert int((exp > 0).sum()) == int((exp[1::4, 1::4] > 0).sum())  # isolated pixels survive alone
    assert set_total > 10000


@pytest.mark.parametrize('variant', ['generic', 'ties'])
def test_fused_mask_other_kernel_variants(env, monkeypatch, variant):
    """The context picks the single-hue-sector kernel for the sample bounds; this
    forces the generic table kernel and the one that re-evaluates rounding ties."""
    from meterelf_amd import MeterReader
    from oracle import pyoracle as po
    monkeypatch.setenv('MELF_FUSED_VARIANT', variant)
    reader = MeterReader(env['sample-images1']['params'])
    try:
        p = reader.ctx.params
        assert reader.ctx.fused_table_ties() == 0
        frames = _blobby(np.random.default_rng(4), 2, 96, 160)
        got = reader.ctx.hls_inrange_close(frames)
        for f in range(2):
            assert np.array_equal(got[f], po.hls_inrange_close(frames[f], p.hue_shift, list(p.needle_lo), list(p.needle_hi)))
    finally:
        reader.close()


@pytest.mark.parametrize('needle', [
    dict(h=125, l=80, s=130, rh=9, rl=45, rs=35, shift=128),    # sample bounds: red sector only
    dict(h=85, l=120, s=120, rh=12, rl=80, rs=100, shift=0),    # green sector
    dict(h=170, l=120, s=120, rh=12, rl=80, rs=100, shift=0),   # blue sector
    dict(h=128, l=128, s=128, rh=100, rl=120, rs=120, shift=0),  # several sectors
    dict(h=43, l=128, s=128, rh=1, rl=128, rs=128, shift=0),    # narrow band at a sector seam (60 deg)
    dict(h=200, l=100, s=200, rh=60, rl=90, rs=55, shift=77),   # odd hue shift, wraps
])
def test_fused_mask_other_bounds(env, tmp_path, needle):
    """Other needle colours select other kernel variants (single g / b sector, generic);
    all must equal the oracle, on random and on blob-structured frames."""
    import shutil
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    src = os.path.join(GOLDEN, 'sample-images1')
    text = open(os.path.join(src, 'params.yml')).read()
    text = text.replace('hue_shift: 128', 'hue_shift: %d' % needle['shift'])
    text = text.replace('needle_color: {h: 125, l: 80, s: 130}', 'needle_color: {h: %(h)d, l: %(l)d, s: %(s)d}' % needle)
    text = text.replace('needle_color_range: {h: 9, l: 45, s: 35}', 'needle_color_range: {h: %(rh)d, l: %(rl)d, s: %(rs)d}' % needle)
    (tmp_path / 'params.yml').write_text(text)
    shutil.copy(os.path.join(src, 'dials_gray.png'), tmp_path / 'dials_gray.png')
    reader = MeterReader(_params.load(str(tmp_path / 'params.yml')))
    try:
        p = reader.ctx.params
        (lo, hi) = (list(p.needle_lo), list(p.needle_hi))
        rng = np.random.default_rng(needle['h'])
        frames = np.concatenate([rng.integers(0, 256, size=(2, 64, 256, 3), dtype=np.uint8), _blobby(rng, 1, 64, 256)])
        got = reader.ctx.hls_inrange_close(frames)
        nz = 0
        for f in range(len(frames)):
            exp = po.hls_inrange_close(frames[f], p.hue_shift, lo, hi)
            assert np.array_equal(got[f], exp), f
            nz += int((exp > 0).sum())
        assert nz > 0
        # the same variant through the work queue (a launch of long runs: 600 frames of 240 x 320), against its pieces and the oracle
        big = rng.integers(0, 256, size=(600, 240, 320, 3), dtype=np.uint8)
        big[::2] = _blobby(rng, 300, 240, 320)
        whole = reader.ctx.hls_inrange_close(big)
        parts = np.concatenate([reader.ctx.hls_inrange_close(big[a:a + 40]) for a in range(0, 600, 40)])
        assert np.array_equal(parts, whole)
        for f in (0, 299, 599):
            assert np.array_equal(whole[f], po.hls_inrange_close(big[f], p.hue_shift, lo, hi)), f
    finally:
        reader.close()


def test_fused_mask_on_fixture_frames(env):
    from meterelf_amd._image import imread_bgr
    from oracle import pyoracle as po
    e = env['sample-images1']
    ctx = e['reader'].ctx
    p = ctx.params
    frames = np.stack([imread_bgr(f) for f in e['files'][5:9]])
    got = ctx.hls_inrange_close(frames)
    for f in range(len(frames)):
        exp = po.hls_inrange_close(frames[f], p.hue_shift, list(p.needle_lo), list(p.needle_hi))
        assert np.array_equal(got[f], exp)


@pytest.mark.parametrize('sd', ['sample-images1', 'sample-images2'])
def test_match_ccoeff_bit_exact(env, sd):
    """Whole float32 correlation map + minMaxLoc on fixture L planes and random images."""
    from meterelf_amd._image import imread_bgr
    from oracle import pyoracle as po
    e = env[sd]
    ctx = e['reader'].ctx
    op = e['oparams']
    tpl = op.load_template()
    imgs = []
    for f in e['files'][:6]:
        crop = po.crop_meter(imread_bgr(f), op)
        imgs.append(po.bgr2hls(crop, op.hue_shift)[:, :, 1])
    rng = np.random.default_rng(5)
    imgs.append(rng.integers(0, 256, size=imgs[0].shape, dtype=np.uint8))
    imgs.append(np.full(imgs[0].shape, 255, np.uint8))  # largest possible sums
    imgs.append(np.zeros(imgs[0].shape, np.uint8))       # all-equal map: first position wins
    imgs = np.stack(imgs)
    mv, mx, my, rmap = ctx.match_ccoeff(imgs, want_map=True)
    for i in range(len(imgs)):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmap[i], emap), i
        assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey), i


def test_match_ccoeff_odd_sizes(env):
    """Images that need several row/column tiles and ragged edges."""
    from oracle import pyoracle as po
    e = env['sample-images1']
    ctx = e['reader'].ctx
    tpl = e['oparams'].load_template()
    rng = np.random.default_rng(11)
    for (rows, cols) in [(119, 188), (120, 189), (119 + 45, 188 + 64), (119 + 90, 188 + 130), (300, 400)]:
        imgs = rng.integers(0, 256, size=(2, rows, cols), dtype=np.uint8)
        mv, mx, my, rmap = ctx.match_ccoeff(imgs, want_map=True)
        for i in range(2):
            (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
            assert np.array_equal(rmap[i], emap), (rows, cols, i)
            assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey)


def _dials_crops(e, files):
    from meterelf_amd._image import imread_bgr
    from oracle import pyoracle as po
    op = e['oparams']
    tpl = op.load_template()
    (th, tw) = tpl.shape
    crops = []
    for f in files:
        hls = po.bgr2hls(po.crop_meter(imread_bgr(f), op), op.hue_shift)
        (_v, x, y, _m) = po.match_ccoeff(hls[:, :, 1], tpl)
        crops.append(hls[y:y + th, x:x + tw])
    return np.stack(crops)


@pytest.mark.parametrize('sd', ['sample-images1', 'sample-images2'])
def test_read_dials_on_fixture_crops(env, sd):
    from oracle import pyoracle as po
    e = env[sd]
    files = _good(e['files'])[:40]
    crops = _dials_crops(e, files)
    recs = e['reader'].ctx.read_dials(crops)
    for i in range(len(crops)):
        o = po.read_dials(crops[i], e['oparams'])
        assert int(recs[i]['status']) == o.status == 0
        assert np.allclose(recs[i]['pos'][:4], list(o.pos)[:4], rtol=0, atol=POS_TOL)
        assert '{:07.3f}'.format(float(recs[i]['value'])) == '{:07.3f}'.format(o.value)


def test_read_dials_stress_random_crops(env):
    """Random and blob-structured HLS crops: many tiny contours, the area <= 100
    branch, unreadable dials and missing contours, all against the oracle."""
    from oracle import pyoracle as po
    e = env['sample-images1']
    ctx = e['reader'].ctx
    (th, tw) = (ctx.params.th, ctx.params.tw)
    rng = np.random.default_rng(99)
    crops = []
    for k in range(96):
        if k % 3 == 0:
            c = rng.integers(0, 256, size=(th, tw, 3), dtype=np.uint8)
        elif k % 3 == 1:  # near-constant colour with noise: big blobs with holes
            base = rng.integers(40, 200, size=3)
            c = np.clip(base[None, None, :] + rng.integers(-40, 40, size=(th, tw, 3)), 0, 255).astype(np.uint8)
        else:  # low-res random colour cells
            cells = rng.integers(0, 256, size=(th // 6 + 1, tw // 6 + 1, 3), dtype=np.uint8)
            c = np.kron(cells, np.ones((6, 6, 1), np.uint8))[:th, :tw]
            c = np.clip(c.astype(np.int64) + rng.integers(-12, 12, size=c.shape), 0, 255).astype(np.uint8)
        crops.append(c)
    # every dial's disk (or three quarters of it) in the dial's own colour: > 1 000 (> 750) needle pixels inside the annulus -- more
    # than the angle phase caches (k_dials RING_CAP = 512): the uncached passes
    for (k, quarter_off) in enumerate((False, True, True)):
        c = np.empty((th, tw, 3), np.uint8)
        c[:] = (200, 30, 40)
        (yy, xx) = np.mgrid[0:th, 0:tw]
        for d in range(int(ctx.params.ndials)):
            (cx, cy) = (ctx.params.dial[d].cx, ctx.params.dial[d].cy)
            inside = (xx - cx) ** 2 + (yy - cy) ** 2 <= 27.0 ** 2
            if quarter_off:
                inside &= ~((xx - cx > 2 + k) & (yy - cy > 2))
            c[inside] = (60 + 20 * d, 120, 150)
        crops.append(c)
    crops = np.stack(crops)
    recs = ctx.read_dials(crops)
    seen = set()
    for i in range(len(crops)):
        o = po.read_dials(crops[i], e['oparams'])
        seen.add(o.status)
        assert int(recs[i]['status']) == o.status, (i, int(recs[i]['status']), o.status)
        if o.status == 0:
            assert np.allclose(recs[i]['pos'][:4], list(o.pos)[:4], rtol=0, atol=POS_TOL), i
            assert abs(float(recs[i]['value']) - o.value) < 1e-8, i
        elif o.status == 2:
            assert int(recs[i]['failed_dial']) == o.failed_dial, i
        elif o.status == 3:
            assert int(recs[i]['unreadable_mask']) == o.unreadable_mask, i
    assert 0 in seen and len(seen) >= 2, seen


# ------------------------------------------------------------ whole path ----

@pytest.mark.parametrize('sd,count', [('sample-images1', 81), ('sample-images2', 223)])
def test_full_path_matches_golden_stdout(env, sd, count, capsys):
    """The reference's own end-to-end test (tests/test_meterelf.py:39-96 and
    integration-tests/test_all_sample_images) run against the HIP path through
    the CLI: string-exact lines, one declared tolerance (match val float)."""
    from meterelf_amd import _main
    e = env[sd]
    with open(os.path.join(GOLDEN, sd + '_stdout.txt')) as fp:
        expected = dict(line.split(': ', 1) for line in fp.read().splitlines())
    assert len(e['files']) == count
    cwd = os.getcwd()
    os.chdir(os.path.join(GOLDEN, sd))
    try:
        _main.main(['meterelf', 'params.yml'] + [os.path.basename(f) for f in e['files']])
    finally:
        os.chdir(cwd)
    captured = capsys.readouterr()
    assert captured.err == ''
    lines = captured.out.splitlines()
    assert len(lines) == count
    bad = []
    for line in lines:
        (name, got) = line.split(': ', 1)
        exp = expected[name]
        if got == exp:
            continue
        if name == NOISY_MATCH_VAL:
            pat = r'UNKNOWN Dials not found \(match val = ([0-9.]+)\)'
            (g, x) = (re.fullmatch(pat, got), re.fullmatch(pat, exp))
            assert g and x and abs(float(g.group(1)) - float(x.group(1))) <= 1e-5 * float(x.group(1))
            continue
        bad.append((name, got, exp))
    assert bad == []


def test_run_as_a_module_like_the_reference_integration_test():
    """integration-tests/test_all_sample_images of the reference: `python3 -m <package> params.yml <all jpg files, sorted>` from
    inside sample-images1, exit code 0, EMPTY stderr, stdout equal to the expected file -- here a fresh interpreter running
    `python -m meterelf_amd` on the GPU (the one declared tolerance: the match value's last float digits in one line)."""
    import subprocess
    d = os.path.join(GOLDEN, 'sample-images1')
    files = sorted(f for f in os.listdir(d) if f.endswith('.jpg'))
    assert len(files) == 81
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    p = subprocess.run([sys.executable, '-m', 'meterelf_amd', 'params.yml'] + files, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stderr == b''
    with open(os.path.join(GOLDEN, 'sample-images1_stdout.txt')) as fp:
        expected = fp.read().splitlines()
    got = p.stdout.decode().splitlines()
    assert len(got) == len(expected) == 81
    for (g, x) in zip(got, expected):
        if g != x:
            assert g.split(': ', 1)[0] == x.split(': ', 1)[0] == NOISY_MATCH_VAL, (g, x)


@pytest.mark.parametrize('sd', ['sample-images1', 'sample-images2'])
def test_full_path_records_match_oracle(env, sd):
    from meterelf_amd._image import imread_bgr
    from oracle import pyoracle as po
    e = env[sd]
    frames = [imread_bgr(f) for f in e['files']]
    recs = e['reader'].read_many(frames)
    for (i, fr) in enumerate(frames):
        o = po.process_frames(fr[None], e['oparams'])[0]
        _compare_records([recs[i]], [o], tag=os.path.basename(e['files'][i]))


@pytest.mark.parametrize('sd', ['sample-images1', 'sample-images2'])
def test_match_paths_agree_dot4_vs_mfma(env, sd, monkeypatch):
    """The VALU (v_dot4) kernel and the MFMA kernel are two formulations of the same exact
    integer correlation: identical float32 maps, and both identical to the oracle."""
    from meterelf_amd import MeterReader
    from oracle import pyoracle as po
    e = env[sd]
    op = e['oparams']
    tpl = op.load_template()
    rng = np.random.default_rng(21)
    (rows, cols) = (250, 250) if sd == 'sample-images1' else (135, 220)
    imgs = rng.integers(0, 256, size=(35, rows, cols), dtype=np.uint8)   # 35 frames: a full and a ragged group of 32
    imgs[3] = 255
    imgs[4] = 0
    (_mv, _mx, _my, map_mfma) = e['reader'].ctx.match_ccoeff(imgs, want_map=True)
    monkeypatch.setenv('MELF_MATCH', 'dot4')
    r2 = MeterReader(e['params'])
    try:
        (mv2, mx2, my2, map_dot4) = r2.ctx.match_ccoeff(imgs, want_map=True)
    finally:
        r2.close()
    assert np.array_equal(map_mfma, map_dot4)
    for i in (0, 3, 4, 34):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(map_dot4[i], emap)
        assert (float(mv2[i]), int(mx2[i]), int(my2[i])) == (ev, ex, ey)


@pytest.mark.parametrize('sd,seed', [('sample-images1', 7), ('sample-images2', 8)])
def test_full_path_dot4_kernel(env, sd, seed, monkeypatch):
    """Whole path with the VALU match kernel forced (the default is the MFMA kernel)."""
    from meterelf_amd import MeterReader
    from oracle import pyoracle as po
    e = env[sd]
    monkeypatch.setenv('MELF_MATCH', 'dot4')
    reader = MeterReader(e['params'])
    try:
        frames = synth_frames(_good(e['files']), 12, seed)
        _compare_records(reader.read_frames(frames), po.process_frames(frames, e['oparams']), tag=sd)
    finally:
        reader.close()


def synth_frames(files, n, seed, shift=8, sigma=2.0):
    """BASELINE config 3/4 synthesis: fixture (i mod K)
    circularly shifted by (dx, dy) in [-shift, shift]^2 plus N(0, sigma^2) noise."""
    from meterelf_amd._image import imread_bgr
    base = [imread_bgr(f) for f in files]
    shape = base[0].shape
    base = [b for b in base if b.shape == shape]
    rng = np.random.default_rng(seed)
    out = np.empty((n,) + shape, np.uint8)
    for i in range(n):
        (dx, dy) = rng.integers(-shift, shift + 1, size=2)
        img = np.roll(base[i % len(base)], (int(dy), int(dx)), axis=(0, 1)).astype(np.int16)
        img += np.rint(rng.normal(0.0, sigma, size=shape)).astype(np.int16)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


@pytest.mark.parametrize('sd,seed', [('sample-images1', 2024), ('sample-images2', 2025)])
def test_full_path_synthetic_batch(env, sd, seed):
    from oracle import pyoracle as po
    e = env[sd]
    good = _good(e['files'])
    frames = synth_frames(good, 48, seed)
    recs = e['reader'].read_frames(frames)
    ores = po.process_frames(frames, e['oparams'])
    _compare_records(recs, ores, tag=sd)
    assert sum(1 for o in ores if o.status == 0) >= 40


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_read_dials_random_geometries(env, tmp_path, seed):
    """Random params.yml dial geometries (centres incl. half-integer ones, diameters, ring
    thickness, colour ranges, momentum sign, zero angle, 3..6 dials) on noisy blob crops:
    the per-dial kernel's windows / bit masks / flood fills against the oracle."""
    import shutil
    import yaml
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)
    src = os.path.join(GOLDEN, 'sample-images1')
    with open(os.path.join(src, 'params.yml')) as fp:
        data = yaml.safe_load(fp)
    ndials = int(rng.integers(3, 7))
    needles = []
    for k in range(ndials):
        diameter = int(rng.integers(6, 20))
        dist = int(rng.integers(1, 6))
        thick = int(rng.integers(3, 11))
        r_out = int(np.rint(diameter / 2.0)) + dist + thick
        cx = float(rng.integers(r_out + 3, 188 - r_out - 3)) + float(rng.choice([0.0, 0.3, 0.5, 0.9]))
        cy = float(rng.integers(r_out + 3, 119 - r_out - 3)) + float(rng.choice([0.0, 0.4, 0.5, 0.7]))
        needles.append({
            'name': 'd%d' % k, 'color_range': {'h': int(rng.integers(5, 40)), 'l': int(rng.integers(20, 90)),
                                               's': int(rng.integers(20, 120))},
            'dist_from_center': dist, 'circle_thickness': thick, 'angle_of_zero': float(rng.uniform(-20, 20)),
            'center': [cx, cy], 'diameter': diameter, 'negative_momentum': bool(rng.integers(0, 2))})
    data['needle_data'] = needles
    with open(tmp_path / 'params.yml', 'w') as fp:
        yaml.safe_dump(data, fp)
    shutil.copy(os.path.join(src, 'dials_gray.png'), tmp_path / 'dials_gray.png')
    params = _params.load(str(tmp_path / 'params.yml'))
    op = po.Params(str(tmp_path / 'params.yml'))
    assert np.array_equal(_hip_masks(params), op.masks())
    crops = []
    for k in range(24):
        base = rng.integers(30, 220, size=3)
        c = np.clip(base[None, None, :] + rng.integers(-35, 35, size=(119, 188, 3)), 0, 255).astype(np.uint8)
        for nd in needles:  # a needle-like streak of a distinct colour through every dial
            (cx, cy) = nd['center']
            ang = rng.uniform(0, 2 * np.pi)
            col = rng.integers(0, 256, size=3)
            for t in np.linspace(0, 26, 80):
                for w in (-1, 0, 1):
                    (x, y) = (int(cx + t * np.cos(ang) + w * np.sin(ang)), int(cy + t * np.sin(ang) - w * np.cos(ang)))
                    if 0 <= x < 188 and 0 <= y < 119:
                        c[y, x] = np.clip(col + rng.integers(-6, 6, size=3), 0, 255)
        crops.append(c)
    crops = np.stack(crops)
    reader = MeterReader(params)
    try:
        recs = reader.ctx.read_dials(crops)
    finally:
        reader.close()
    statuses = set()
    for i in range(len(crops)):
        o = po.read_dials(crops[i], op)
        statuses.add(o.status)
        assert int(recs[i]['status']) == o.status, (i, int(recs[i]['status']), o.status)
        if o.status == 0:
            assert np.allclose(recs[i]['pos'][:ndials], list(o.pos)[:ndials], rtol=0, atol=POS_TOL), i
            if ndials == 4:
                assert abs(float(recs[i]['value']) - o.value) < 1e-8
        elif o.status == 2:
            assert int(recs[i]['failed_dial']) == o.failed_dial
        elif o.status == 3:
            assert int(recs[i]['unreadable_mask']) == o.unreadable_mask
    assert 0 in statuses


@pytest.mark.parametrize('tw,th,kind', [(188, 119, 'random'), (188, 119, 'extreme'), (170, 100, 'random'), (150, 90, 'random'), (64, 40, 'random')])
def test_match_other_templates(tmp_path, tw, th, kind):
    """Other templates: random and 0/255-extreme content (limits of the T-128 / L-128 int8 trick) at
    the MFMA kernel's width class, and other widths that take the generic v_dot4 kernel."""
    import shutil
    import yaml
    from PIL import Image
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    rng = np.random.default_rng(tw * 1000 + th + len(kind))
    src = os.path.join(GOLDEN, 'sample-images1')
    with open(os.path.join(src, 'params.yml')) as fp:
        data = yaml.safe_load(fp)
    data['dials_template_size'] = [tw, th]
    for nd in data['needle_data']:   # keep the dial windows inside the smaller templates
        nd['center'] = [min(nd['center'][0], tw - 26.0), min(nd['center'][1], th - 26.0)]
        nd['center'] = [max(nd['center'][0], 26.0), max(nd['center'][1], 26.0)]
    with open(tmp_path / 'params.yml', 'w') as fp:
        yaml.safe_dump(data, fp)
    if kind == 'extreme':
        tpl = rng.choice(np.array([0, 255], np.uint8), size=(th, tw))
    else:
        tpl = rng.integers(0, 256, size=(th, tw), dtype=np.uint8)
    Image.fromarray(tpl, 'L').save(tmp_path / 'dials_gray.png')
    params = _params.load(str(tmp_path / 'params.yml'))
    imgs = rng.integers(0, 256, size=(5, 250, 250), dtype=np.uint8)
    imgs[1] = 255
    imgs[2] = rng.choice(np.array([0, 255], np.uint8), size=(250, 250))
    imgs[3, 40:40 + th, 30:30 + tw] = tpl   # an exact occurrence of the template
    reader = MeterReader(params)
    try:
        (mv, mx, my, rmap) = reader.ctx.match_ccoeff(imgs, want_map=True)
    finally:
        reader.close()
    for i in range(len(imgs)):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmap[i], emap), i
        assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey), i
    assert (int(mx[3]), int(my[3])) == (30, 40)


def _hip_masks(params):
    from meterelf_amd import _hip
    return _hip.build_dial_masks(params.to_c())


def test_1080p_six_dials(env, tmp_path):
    """BASELINE config 5 shape: 1920x1080 frames, meter_rect inside the frame, six dials.
    The reference cannot compute `value` for != 4 dials (assert in _reading.py:166): the record
    carries the six positions only.  Oracle parity on positions / match / statuses."""
    import shutil
    import yaml
    from meterelf_amd import MeterReader, _params
    from meterelf_amd._image import imread_bgr
    from oracle import pyoracle as po
    src = os.path.join(GOLDEN, 'sample-images1')
    with open(os.path.join(src, 'params.yml')) as fp:
        data = yaml.safe_load(fp)
    data['meter_rect'] = {'top_left': [1210, 420], 'bottom_right': [1460, 670]}
    extra = []
    for (k, nd) in enumerate(data['needle_data'][:2]):
        nd2 = dict(nd)
        nd2['name'] = '1.%d' % k
        nd2['center'] = [nd['center'][0] + 0.4, nd['center'][1] - 0.3]
        extra.append(nd2)
    data['needle_data'] = data['needle_data'] + extra
    with open(tmp_path / 'params.yml', 'w') as fp:
        yaml.safe_dump(data, fp)
    shutil.copy(os.path.join(src, 'dials_gray.png'), tmp_path / 'dials_gray.png')
    params = _params.load(str(tmp_path / 'params.yml'))
    op = po.Params(str(tmp_path / 'params.yml'))
    assert len(params.dial_names) == 6
    rng = np.random.default_rng(1080)
    files = _good(env['sample-images1']['files'])[:5]
    frames = rng.integers(0, 256, size=(len(files), 1080, 1920, 3), dtype=np.uint8)
    for (i, f) in enumerate(files):
        crop = imread_bgr(f)[160:410, 50:300]
        frames[i, 420:670, 1210:1460] = crop
    reader = MeterReader(params)
    try:
        recs = reader.read_frames(frames)
        ores = po.process_frames(frames, op)
        for i in range(len(files)):
            (r, o) = (recs[i], ores[i])
            assert int(r['status']) == o.status == 0
            assert (int(r['match_x']), int(r['match_y']), float(r['match_val'])) == (o.match_x, o.match_y, o.match_val)
            assert np.allclose(r['pos'][:6], list(o.pos)[:6], rtol=0, atol=POS_TOL)
        # the fused full-frame stage at 1080p
        got = reader.ctx.hls_inrange_close(frames[:2])
        p = reader.ctx.params
        for f in range(2):
            assert np.array_equal(got[f], po.hls_inrange_close(frames[f], p.hue_shift, list(p.needle_lo), list(p.needle_hi)))
    finally:
        reader.close()


def test_large_and_ragged_batches(env):
    """Batch sizes around the 32-frame MFMA group and a batch of 70 (two groups + 6)."""
    from oracle import pyoracle as po
    e = env['sample-images2']
    frames = synth_frames(_good(e['files']), 70, 31)
    ores = po.process_frames(frames, e['oparams'])
    for n in (1, 31, 32, 33, 70):
        recs = e['reader'].read_frames(frames[:n])
        _compare_records(recs, [ores[i] for i in range(n)], tag='n=%d' % n)


def test_edge_cases(env):
    from meterelf_amd import _hip
    e = env['sample-images1']
    reader = e['reader']
    # empty batch
    assert len(reader.read_frames(np.zeros((0, 640, 480, 3), np.uint8))) == 0
    # constant frame: correlation map is identically 0 -> first position, below threshold
    recs = reader.read_frames(np.full((2, 640, 480, 3), 128, np.uint8))
    assert [int(r['status']) for r in recs] == [1, 1]
    assert [(int(r['match_x']), int(r['match_y']), float(r['match_val'])) for r in recs] == [(0, 0, 0.0)] * 2
    # frame too small for the template inside meter_rect: the C ABI refuses, loudly
    with pytest.raises(_hip.HipError):
        reader.read_frames(np.zeros((1, 200, 200, 3), np.uint8))
    # pre-cropped injection (ImageFile.bgr_image semantics)
    from meterelf_amd._image import ImageFile, imread_bgr
    from meterelf_amd._reading import get_meter_value
    from meterelf_amd._engine import _readers
    p = e['params']
    f = [x for x in e['files'] if x.endswith('e136.jpg')][0]
    ((x0, y0), (x1, y1)) = p.meter_rect
    crop = np.ascontiguousarray(imread_bgr(f)[y0:y1, x0:x1])
    vals = get_meter_value(ImageFile(f, p, bgr_image=crop))
    assert abs(vals['value'] - 253.62306) < 0.000005  # reference tests/test_meterelf.py:187
    assert abs(vals['0.0001'] - 6.23) < 0.005
    vals2 = get_meter_value(ImageFile(f, p))
    assert vals2 == vals
    _readers.pop(id(p)).close()


def test_stream_of_batches_matches_single_calls(env):
    """melf_process_stream_dev (batches alternating between the two pipeline lanes) gives the records of one
    melf_process_batch call per batch."""
    import ctypes as C
    from meterelf_amd import _hip
    e = env['sample-images1']
    frames = synth_frames(_good(e['files']), 96, 31)
    ctx = e['reader'].ctx
    ref = ctx.process_batch(frames)
    L = _hip.lib()
    hip = hip_runtime()
    nbytes = frames.nbytes
    (d_frames, d_res) = (C.c_void_p(), C.c_void_p())
    assert hip.hipMalloc(C.byref(d_frames), C.c_size_t(nbytes)) == 0
    assert hip.hipMalloc(C.byref(d_res), C.c_size_t(len(frames) * _hip.RESULT_DTYPE.itemsize)) == 0
    try:
        assert hip.hipMemcpy(d_frames, frames.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), 1) == 0
        (n, H, W) = (32, frames.shape[1], frames.shape[2])
        ctx.process_stream_dev(d_frames.value, 3, n * H * W * 3, n, H, W, d_res.value, n)
        assert hip.hipDeviceSynchronize() == 0
        got = np.zeros(len(frames), _hip.RESULT_DTYPE)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d_res, C.c_size_t(got.nbytes), 2) == 0
        assert got.tobytes() == ref.tobytes()
    finally:
        hip.hipFree(d_frames)
        hip.hipFree(d_res)


def test_host_fed_batches_crop_upload(env):
    """melf_process_batch uploads only the meter_rect crop, in 256-frame chunks through two pinned staging buffers:
    600 frames = three chunks (the third re-uses the first buffer), a ragged last chunk, against the oracle and against
    the same frames processed from HBM (melf_process_batch_dev on whole frames)."""
    import ctypes as C
    from meterelf_amd import _hip
    from oracle import pyoracle as po
    e = env['sample-images2']
    frames = synth_frames(_good(e['files']), 600, 77)
    ctx = e['reader'].ctx
    recs = ctx.process_batch(frames)
    ores = po.process_frames(frames, e['oparams'])
    _compare_records(recs, ores, tag='host-fed')
    assert sum(int(r['status']) == 0 for r in recs) > 550
    hip = hip_runtime()
    d_frames = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_frames), C.c_size_t(frames.nbytes)) == 0
    try:
        assert hip.hipMemcpy(d_frames, frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
        (n, H, W) = frames.shape[:3]
        dev = ctx.process_batch_dev(d_frames.value, n, H, W)
        assert dev.tobytes() == recs.tobytes()
    finally:
        hip.hipFree(d_frames)
    # a second, smaller call re-uses the staging buffers
    again = ctx.process_batch(frames[5:40])
    assert again.tobytes() == recs[5:40].tobytes()


def _reader_with_template(tmp_path, tpl):
    """A MeterReader whose dials template is `tpl` (the match stage does not look at the dials)."""
    import yaml
    from PIL import Image
    from meterelf_amd import MeterReader, _params
    (th, tw) = tpl.shape
    with open(os.path.join(GOLDEN, 'sample-images1', 'params.yml')) as fp:
        data = yaml.safe_load(fp)
    data['dials_template_size'] = [tw, th]
    small = min(th, tw) < 52
    for nd in data['needle_data']:
        if small:   # dials that fit a tiny template: radius round(2 / 2) + 0 + 2 - 1 = 2
            (nd['diameter'], nd['dist_from_center'], nd['circle_thickness']) = (2, 0, 2)
            nd['center'] = [tw / 2.0, th / 2.0]
        else:
            nd['center'] = [max(min(nd['center'][0], tw - 26.0), 26.0), max(min(nd['center'][1], th - 26.0), 26.0)]
    tmp_path.mkdir(parents=True, exist_ok=True)
    with open(tmp_path / 'params.yml', 'w') as fp:
        yaml.safe_dump(data, fp)
    Image.fromarray(tpl, 'L').save(tmp_path / 'dials_gray.png')
    return MeterReader(_params.load(str(tmp_path / 'params.yml')))


GEN_SHAPES = [
    # (th, tw, rows, cols, n): what the case exercises
    (119, 188, 135, 220, 35),   # config 4: 17 x 33 map = one column block + one V-form column, K slices
    (119, 188, 250, 250, 3),    # config 3 shape at a small batch: sliced tiles
    (119, 188, 300, 400, 2),    # 182 x 213 map: seven column blocks in strips of two, V-form off (remainder 21)
    (119, 188, 300, 223, 2),    # 182 x 36 map: V form with 4 remainder columns and 6 row blocks
    (100, 170, 250, 250, 5),    # other template widths (6 Toeplitz blocks) ...
    (90, 150, 250, 250, 5),
    (40, 64, 250, 250, 33),     # ... a narrow one (3 blocks), 211 x 187 map, two frame groups
    (119, 256, 200, 300, 2),    # the widest template the u16 row sums take (9 blocks)
    (9, 11, 64, 70, 2),         # tiny template: two Toeplitz blocks per row
    (119, 188, 119, 188, 2),    # 1 x 1 map
    (33, 33, 96, 97, 4),        # 64 x 65: two column blocks + 1 V column, rows 64
]


@pytest.mark.parametrize('shape', GEN_SHAPES, ids=lambda s: 't%dx%d_i%dx%d_n%d' % s)
def test_match_general_matrix_core_kernel(tmp_path, monkeypatch, shape):
    """k_match_gen (any template up to 256 columns, any map, K slices, V-form remainder columns): the whole float32
    map and minMaxLoc bit-exact against the oracle, and identical to the VALU kernel's."""
    from oracle import pyoracle as po
    (th, tw, rows, cols, n) = shape
    rng = np.random.default_rng(th * 7 + tw * 3 + rows + cols + n)
    tpl = rng.integers(0, 256, size=(th, tw), dtype=np.uint8)
    imgs = rng.integers(0, 256, size=(n, rows, cols), dtype=np.uint8)
    imgs[-1] = rng.choice(np.array([0, 255], np.uint8), size=(rows, cols))
    if rows >= th + 5 and cols >= tw + 3:
        imgs[0, 5:5 + th, 3:3 + tw] = tpl   # an exact occurrence
    monkeypatch.setenv('MELF_MATCH', 'gen')
    reader = _reader_with_template(tmp_path / 'gen', tpl)
    try:
        (mv, mx, my, rmap) = reader.ctx.match_ccoeff(imgs, want_map=True)
        (mv2, mx2, my2, _none) = reader.ctx.match_ccoeff(imgs, want_map=False)   # a second launch: counters were reset
    finally:
        reader.close()
    assert (mv.tobytes(), mx.tobytes(), my.tobytes()) == (mv2.tobytes(), mx2.tobytes(), my2.tobytes())
    monkeypatch.setenv('MELF_MATCH', 'dot4')
    r2 = _reader_with_template(tmp_path / 'dot4', tpl)
    try:
        (mvd, mxd, myd, rmapd) = r2.ctx.match_ccoeff(imgs, want_map=True)
    finally:
        r2.close()
    assert np.array_equal(rmap, rmapd)
    for i in range(n):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmap[i], emap), i
        assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey), i
        assert (float(mvd[i]), int(mxd[i]), int(myd[i])) == (ev, ex, ey), i
    if rows >= th + 5 and cols >= tw + 3:
        assert (int(mx[0]), int(my[0])) == (3, 5)


@pytest.mark.parametrize('sd,kind', [('sample-images1', 'fast'), ('sample-images1', 'gen'), ('sample-images2', 'fast'), ('sample-images2', 'gen')])
def test_full_path_with_either_matrix_core_kernel(env, monkeypatch, sd, kind):
    """The whole path with the tuned and with the general matrix-core kernel forced: identical records, equal to the oracle."""
    from meterelf_amd import MeterReader
    from oracle import pyoracle as po
    e = env[sd]
    frames = synth_frames(_good(e['files']), 70, 99)
    ores = po.process_frames(frames, e['oparams'])
    monkeypatch.setenv('MELF_MATCH', kind)
    r = MeterReader(e['params'])
    try:
        recs = r.read_frames(frames)
        recs2 = r.read_frames(frames[:33])
    finally:
        r.close()
    _compare_records(recs, ores, tag=kind)
    assert recs2.tobytes() == recs[:33].tobytes()


def test_ctx_create_bcast_one_device(env):
    """melf_ctx_create_bcast with the one GPU a box has: RCCL is bound (dlopen), a one-rank communicator broadcasts the blob in
    place, and the context built from the device-resident bytes reads frames exactly like one built from the host blob."""
    from meterelf_amd import _engine, _hip
    e = env['sample-images1']
    blob = _engine.make_blob(e['params'])
    (ctx,) = _hip.Context.create_bcast(blob, [0])
    try:
        from meterelf_amd._image import imread_bgr
        frames = np.stack([imread_bgr(f) for f in e['files'][2:14]])
        assert ctx.process_batch(frames).tobytes() == e['reader'].ctx.process_batch(frames).tobytes()
        assert np.array_equal(ctx.masks(), e['reader'].ctx.masks())
    finally:
        ctx.close()


def test_two_caller_streams_use_two_lanes(env):
    """A context gives each caller stream its own lane: two batches enqueued on two streams run concurrently and give
    the records of serial calls; a third stream has to take a lane over (ordered by an event) and is right too."""
    import ctypes as C
    from meterelf_amd import _hip
    e = env['sample-images1']
    frames = synth_frames(_good(e['files']), 192, 123)
    ctx = e['reader'].ctx
    ref = ctx.process_batch(frames)
    hip = hip_runtime()
    (n, H, W) = (64, frames.shape[1], frames.shape[2])
    rsz = _hip.RESULT_DTYPE.itemsize
    (d_frames, d_res) = (C.c_void_p(), C.c_void_p())
    streams = [C.c_void_p() for _ in range(3)]
    assert hip.hipMalloc(C.byref(d_frames), C.c_size_t(frames.nbytes)) == 0
    assert hip.hipMalloc(C.byref(d_res), C.c_size_t(len(frames) * rsz)) == 0
    for s in streams:
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0   # hipStreamNonBlocking
    try:
        assert hip.hipMemcpy(d_frames, frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
        for rep in range(3):       # several rounds: lanes are re-used, the third stream keeps taking one over
            assert hip.hipMemset(d_res, 0, C.c_size_t(len(frames) * rsz)) == 0
            assert hip.hipDeviceSynchronize() == 0
            for b in range(3):
                ctx.process_batch_dev(d_frames.value + b * n * H * W * 3, n, H, W, d_results_ptr=d_res.value + b * n * rsz,
                                      want_host=False, stream=streams[(b + rep) % 3].value)
            assert hip.hipDeviceSynchronize() == 0
            got = np.zeros(len(frames), _hip.RESULT_DTYPE)
            assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d_res, C.c_size_t(got.nbytes), 2) == 0
            assert got.tobytes() == ref.tobytes(), rep
        # the null stream (stream = NULL) is a stream like any other
        assert hip.hipMemset(d_res, 0, C.c_size_t(n * rsz)) == 0
        out = ctx.process_batch_dev(d_frames.value, n, H, W, want_host=True, stream=None)
        assert out.tobytes() == ref[:n].tobytes()
    finally:
        ctx.sync()   # the context forgets the streams before they are destroyed
        for s in streams:
            hip.hipStreamDestroy(s)
        hip.hipFree(d_frames)
        hip.hipFree(d_res)


def test_frames_resident_hint_same_records(env):
    """melf_ctx_set_frames_resident: a call's prep kernels run on the lane's side stream under the previous call's dials
    kernel.  Six calls back to back over three resident batches (and a lane change in between): records identical."""
    import ctypes as C
    from meterelf_amd import _hip
    e = env['sample-images2']
    frames = synth_frames(_good(e['files']), 3 * 160, 321)
    ctx = e['reader'].ctx
    ref = ctx.process_batch(frames)
    hip = hip_runtime()
    (n, H, W) = (160, frames.shape[1], frames.shape[2])
    rsz = _hip.RESULT_DTYPE.itemsize
    (d_frames, d_res) = (C.c_void_p(), C.c_void_p())
    streams = [C.c_void_p() for _ in range(2)]
    assert hip.hipMalloc(C.byref(d_frames), C.c_size_t(frames.nbytes)) == 0
    assert hip.hipMalloc(C.byref(d_res), C.c_size_t(2 * len(frames) * rsz)) == 0
    for s in streams:
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    try:
        assert hip.hipMemcpy(d_frames, frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
        assert hip.hipMemset(d_res, 0, C.c_size_t(2 * len(frames) * rsz)) == 0
        assert hip.hipDeviceSynchronize() == 0   # the promise: frames complete before the calls
        ctx.set_frames_resident(True)
        for i in range(6):
            b = i % 3
            st = streams[0] if i != 3 else streams[1]   # call 3 arrives on another stream (takes the other lane)
            ctx.process_batch_dev(d_frames.value + b * n * H * W * 3, n, H, W, d_results_ptr=d_res.value + i * n * rsz,
                                  want_host=False, stream=st.value)
        assert hip.hipDeviceSynchronize() == 0
        got = np.zeros(2 * len(frames), _hip.RESULT_DTYPE)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d_res, C.c_size_t(got.nbytes), 2) == 0
        assert got[:len(frames)].tobytes() == ref.tobytes() and got[len(frames):].tobytes() == ref.tobytes()
    finally:
        ctx.set_frames_resident(False)
        ctx.sync()
        for s in streams:
            hip.hipStreamDestroy(s)
        hip.hipFree(d_frames)
        hip.hipFree(d_res)


def test_frames_resident_hint_with_host_records_and_mode_switches(env):
    """The hint with records returned to the host (each call synchronises), switched on and off between calls, and a
    non-resident call on a caller stream in between (lane hand-over between a lane's own stream and a caller stream)."""
    import ctypes as C
    e = env['sample-images1']
    frames = synth_frames(_good(e['files']), 2 * 320, 555)
    ctx = e['reader'].ctx
    ref = ctx.process_batch(frames)
    hip = hip_runtime()
    (n, H, W) = (320, frames.shape[1], frames.shape[2])
    d_frames = C.c_void_p()
    st = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_frames), C.c_size_t(frames.nbytes)) == 0
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
    try:
        assert hip.hipMemcpy(d_frames, frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
        assert hip.hipDeviceSynchronize() == 0
        for (i, hint) in enumerate([True, True, False, True, True, True, False, False]):
            ctx.set_frames_resident(hint)
            b = i % 2
            got = ctx.process_batch_dev(d_frames.value + b * n * H * W * 3, n, H, W, want_host=True, stream=st.value if i % 3 else None)
            assert got.tobytes() == ref[b * n:(b + 1) * n].tobytes(), (i, hint)
    finally:
        ctx.set_frames_resident(False)
        ctx.sync()
        hip.hipStreamDestroy(st)
        hip.hipFree(d_frames)


# ---- BASELINE.json's full per-GPU sizes: the oracle is too slow there, so what is checked is what cannot depend on the
# ---- batch -- a frame's record is a function of that frame alone -- plus the oracle on a sample of the same frames
def _records_equal(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize('sd,seed', [('sample-images1', 2024), ('sample-images2', 2025)])
def test_full_size_batch_properties(env, sd, seed):
    """Configs 3 and 4: 1024 frames in one call -- host-fed (melf_process_batch: 128-frame chunks through the general kernel)
    AND resident in HBM (ONE melf_process_batch_dev launch of 1024 frames: the tuned kernel's headline layout for config 3,
    the general kernel's 32-group plan for config 4 -- the launches bench.py times).  (0) Both give the same records.
    (1) Batch-composition invariance: the records equal those of the same frames in ragged pieces (one piece exercises the
    small-batch kernels).  (2) Permutation equivariance.  (3) A frame repeated in the batch gives the same record everywhere.
    (4) The oracle on 48 of the 1024."""
    import ctypes as C
    from oracle import pyoracle as po
    e = env[sd]
    reader = e['reader']
    base = synth_frames(_good(e['files']), 256, seed, shift=12, sigma=6.0)
    rng = np.random.default_rng(seed + 1)
    pick = rng.integers(0, 256, 1024)
    pick[:256] = np.arange(256)
    frames = base[pick]  # every base frame at least once, the rest repeats in random places
    whole = reader.read_frames(frames)
    assert len(whole) == 1024
    # (0) the same 1024 frames resident in HBM, one call, default dispatch
    hip = hip_runtime()
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(frames.nbytes)) == 0
    try:
        assert hip.hipMemcpy(dptr, frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
        resident = reader.ctx.process_batch_dev(dptr.value, 1024, frames.shape[1], frames.shape[2])
        info = reader.ctx.last_match()
    finally:
        hip.hipFree(dptr)
    want = {'sample-images1': ('mfma', 'rb4+pairs', 1024), 'sample-images2': ('gen', 'r4x1/4+v1', 1024)}[sd]
    assert (info['kernel'], info['layout'], info['n']) == want, info
    assert _records_equal(resident, whole), 'one resident 1024-frame launch vs the host-fed 128-frame chunks'
    # (1) pieces: 1000 + 24, and 33 + 479 + 512
    for cuts in ((0, 1000, 1024), (0, 33, 512, 1024)):
        parts = np.concatenate([reader.read_frames(frames[a:b]) for (a, b) in zip(cuts, cuts[1:])])
        assert _records_equal(parts, whole), cuts
    # (2) a permutation of the batch permutes the records
    perm = rng.permutation(1024)
    assert _records_equal(reader.read_frames(frames[perm]), whole[perm])
    # (3) repeats
    first = {}
    for (i, b) in enumerate(pick.tolist()):
        j = first.setdefault(b, i)
        assert whole[i].tobytes() == whole[j].tobytes(), (i, j)
    # (4) the oracle on a sample
    sample = rng.choice(1024, 48, replace=False)
    _compare_records(whole[sample], po.process_frames(frames[sample], e['oparams']), tag=sd + ' full size')
    statuses = set(int(s) for s in whole['status'])
    assert 0 in statuses and len(statuses) >= 2  # readable frames and at least one kind of failure in the batch


@pytest.mark.parametrize('n,H,W', [(64, 1080, 1920), (480, 480, 640), (72, 1077, 1904)])
def test_fused_mask_work_queue_launches(env, n, H, W):
    """Launches whose workgroups would each stream a long run of rows take small segments from a work queue instead of the
    static split (k_hls.hip, launch_lut_t: six or more segments of three passes per workgroup).  The same frames in
    pieces small enough for the static split give the same masks, as does the oracle on a sample; every segment border
    (20 or 71 rows apart) lies inside a frame, so a wrong halo shows up as a wrong row there."""
    from oracle import pyoracle as po
    ctx = env['sample-images1']['reader'].ctx
    p = ctx.params
    rng = np.random.default_rng(n + H)
    tiles = _blobby(rng, 6, H, W)
    frames = tiles[rng.integers(0, 6, n)]
    for i in range(n):
        frames[i] = np.roll(frames[i], (i * 13) % H, axis=0)
    whole = ctx.hls_inrange_close(frames)
    assert whole.shape == (n, H, W) and set(np.unique(whole).tolist()) <= {0, 255} and (whole > 0).any()
    step = max(1, n // 16)
    parts = np.concatenate([ctx.hls_inrange_close(frames[a:a + step]) for a in range(0, n, step)])
    assert np.array_equal(parts, whole)
    lo, hi = list(p.needle_lo), list(p.needle_hi)
    for f in rng.choice(n, 3, replace=False):
        exp = po.hls_inrange_close(frames[f], p.hue_shift, lo, hi)
        assert np.array_equal(whole[f], exp), (f, np.argwhere(whole[f] != exp)[:5])
    # a second launch right behind the first finds its queue slot zeroed again
    assert np.array_equal(ctx.hls_inrange_close(frames), whole)


def test_full_size_fused_mask_properties(env):
    """Config 2: B = 256 frames of 640 x 480 in one launch.  The mask of a frame does not depend on its neighbours in the
    batch (pieces, permutation), closing is idempotent on its own output's domain (mask pixels are 0 / 255 only), and
    the oracle agrees on 6 of the 256 frames."""
    from oracle import pyoracle as po
    ctx = env['sample-images1']['reader'].ctx
    p = ctx.params
    rng = np.random.default_rng(1234)
    tiles = _blobby(rng, 16, 480, 640)
    pick = rng.integers(0, 16, 256)
    frames = tiles[pick]
    # per-frame variation so that no two frames are equal: a different circular shift each
    for i in range(256):
        frames[i] = np.roll(frames[i], (i * 7) % 480, axis=0)
    whole = ctx.hls_inrange_close(frames)
    assert whole.shape == (256, 480, 640) and set(np.unique(whole).tolist()) <= {0, 255}
    parts = np.concatenate([ctx.hls_inrange_close(frames[a:b]) for (a, b) in ((0, 100), (100, 101), (101, 256))])
    assert np.array_equal(parts, whole)
    perm = rng.permutation(256)
    assert np.array_equal(ctx.hls_inrange_close(frames[perm]), whole[perm])
    lo, hi = list(p.needle_lo), list(p.needle_hi)
    for f in rng.choice(256, 6, replace=False):
        assert np.array_equal(whole[f], po.hls_inrange_close(frames[f], p.hue_shift, lo, hi)), f
    assert (whole > 0).any()


@pytest.mark.gpu
def test_dial_window_fetch_edge_paths(env, tmp_path):
    """k_dials fetches a dial window as aligned 16-byte pieces (four pixels per lane) when the window lies inside the
    crop's columns and the frames' buffer is 4-byte aligned, and pixel by pixel through the exact path otherwise.
    (a) the same frames at device addresses 1, 2 and 3 bytes off alignment: the aligned run's records byte for byte, which
    equal the oracle's; (b) dials whose windows leave the crop on the left and on the right: against the oracle."""
    import ctypes as C
    import shutil
    import yaml
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    e = env['sample-images1']
    frames = synth_frames(_good(e['files']), 96, 5)
    (n, H, W) = frames.shape[:3]
    ctx = e['reader'].ctx
    hip = hip_runtime()
    d_buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_buf), C.c_size_t(frames.nbytes + 8)) == 0
    try:
        ref = None
        for off in (0, 1, 2, 3):
            assert hip.hipMemcpy(C.c_void_p(d_buf.value + off), frames.ctypes.data_as(C.c_void_p), C.c_size_t(frames.nbytes), 1) == 0
            got = ctx.process_batch_dev(d_buf.value + off, n, H, W)
            if ref is None:
                ref = got
                _compare_records(ref, po.process_frames(frames, e['oparams']), tag='aligned')
                assert sum(int(r['status']) == 0 for r in ref) > 80
            else:
                assert got.tobytes() == ref.tobytes(), 'frames %d byte(s) off alignment' % off
    finally:
        hip.hipFree(d_buf)
    # (b)
    with open(e['pfile']) as fp:
        data = yaml.safe_load(fp)
    tw = data['dials_template_size'][0]
    data['needle_data'][0]['center'][0] = 18.0        # window columns -5 .. 41
    data['needle_data'][3]['center'][0] = tw - 17.5   # ... and beyond the crop's last column
    shutil.copy(os.path.join(GOLDEN, 'sample-images1', data['dials_template']), tmp_path / data['dials_template'])
    with open(tmp_path / 'params.yml', 'w') as fp:
        yaml.safe_dump(data, fp)
    reader = MeterReader(_params.load(str(tmp_path / 'params.yml')))
    try:
        recs = reader.read_frames(frames[:48])
        _compare_records(recs, po.process_frames(frames[:48], po.Params(str(tmp_path / 'params.yml'))), tag='windows beyond the crop')
    finally:
        reader.close()


@pytest.mark.gpu
def test_ring_points_beyond_both_angle_caches(env, tmp_path):
    """k_dials caches the angles of the needle's pixels inside the annulus in LDS (<= 512 points), in registers (<= 1024) or
    not at all (beyond: an arctangent per point and pass).  A dial with a 20-pixel annulus, its disk wholly / three quarters /
    one quarter in the dial's colour: ~2 100 / ~1 600 / ~500 ring points -- all three tiers against the oracle."""
    import shutil
    import yaml
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    e = env['sample-images1']
    with open(e['pfile']) as fp:
        data = yaml.safe_load(fp)
    nd0 = data['needle_data'][0]
    (nd0['diameter'], nd0['dist_from_center'], nd0['circle_thickness']) = (10, 2, 20)
    shutil.copy(os.path.join(GOLDEN, 'sample-images1', data['dials_template']), tmp_path / data['dials_template'])
    with open(tmp_path / 'params.yml', 'w') as fp:
        yaml.safe_dump(data, fp)
    reader = MeterReader(_params.load(str(tmp_path / 'params.yml')))
    oparams = po.Params(str(tmp_path / 'params.yml'))
    try:
        ctx = reader.ctx
        (th, tw) = (ctx.params.th, ctx.params.tw)
        (yy, xx) = np.mgrid[0:th, 0:tw]
        crops = []
        for kind in range(6):
            c = np.empty((th, tw, 3), np.uint8)
            c[:] = (200, 30, 40)
            for d in range(int(ctx.params.ndials)):
                (cx, cy) = (ctx.params.dial[d].cx, ctx.params.dial[d].cy)
                inside = (xx - cx) ** 2 + (yy - cy) ** 2 <= 30.0 ** 2
                if kind % 3 == 1:
                    inside &= ~((xx - cx > 2) & (yy - cy > 3))
                if kind % 3 == 2:
                    inside &= (xx - cx < -1) & (yy - cy > 1 + kind)
                c[inside] = (60 + 20 * d, 120, 150)
            if kind >= 3:   # ... and with noise in the lightness, so that the masks are ragged
                rng = np.random.default_rng(kind)
                c[..., 1] = np.clip(c[..., 1].astype(np.int64) + rng.integers(-30, 30, size=(th, tw)), 0, 255).astype(np.uint8)
            crops.append(c)
        crops = np.stack(crops)
        recs = ctx.read_dials(crops)
        for i in range(len(crops)):
            o = po.read_dials(crops[i], oparams)
            assert int(recs[i]['status']) == o.status, (i, int(recs[i]['status']), o.status)
            if o.status == 0:
                assert np.allclose(recs[i]['pos'][:4], list(o.pos)[:4], rtol=0, atol=POS_TOL), i
                assert np.allclose(recs[i]['angle'][:4], list(o.angle)[:4], rtol=0, atol=POS_TOL), i
            elif o.status == 3:
                assert int(recs[i]['unreadable_mask']) == o.unreadable_mask, i
    finally:
        reader.close()
