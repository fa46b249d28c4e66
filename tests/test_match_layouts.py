"""The tuned matrix-core match kernel (k_match_mfma) in EVERY wave layout its planner can pick, at default dispatch.

The reference has one code path for every batch size (cv2.matchTemplate + minMaxLoc, meterelf/_utils.py:91-97, called
per image from meterelf/_image.py:57-66).  Here the batch size selects the kernel (pick_match_kind) and, for the tuned
kernel, the wave layout (mfma_plan: RB = 2..5 full map rows per wave, with or without pairs of (RB + 1)-row waves that
share a map row).  These tests launch each of them the way production does -- frames resident in HBM, one
melf_process_batch_dev call, no MELF_MATCH override -- and ASSERT which kernel and layout ran (melf_ctx_last_match), so
that a change of a dispatch threshold cannot silently move a test onto another kernel.

CPU part (not gpu): the planner's invariants over every batch size 1..4300.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tests.helpers import hip_runtime

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
REJECTED = ('20180814021309-01-e01.jpg', '20180814021310-00-e02.jpg')

# batch size -> layout the planner must pick for the sample-images1 shape (crop 250 x 250, template 119 x 188, map
# 132 x 63) on 1024 SIMDs.  128 / 256 (the host-fed chunk size) / 320 / 384: the tuned kernel's small size classes at default
# dispatch (four / two K slices; below 500 waves -- up to 96 frames -- the general kernel); 512: an 8-GPU shard of config 5;
# 600 ... 900: mid-size batches -- round 6: 4-row waves on part of the SIMDs (the chip is power-limited: fewer, larger tiles beat
# 2- and 3-row layouts that fill every SIMD; rounds 3-5 picked rb2+pairs / rb3 / rb3+pairs here); 1024: the BENCH headline launch
# and get_meter_values' default chunk; 1056: 33 groups; 1100: 5-row waves; 2048: two full rounds of waves.
# "/kN" (round 5): N waves per tile, each over 1 / N of the template rows, adding up in the workgroup's LDS -- the 4-row tiles of
# the 1024-frame layout for batches that would otherwise run 2-row waves (320, 512: config 5's per-GPU share).
EXPECTED_LAYOUT = {128: 'rb4/k4', 256: 'rb4+pairs/k4', 320: 'rb4/k2', 384: 'rb4/k2', 512: 'rb4+pairs/k2', 600: 'rb4', 800: 'rb4', 900: 'rb4', 1024: 'rb4+pairs',
                   1056: 'rb4+pairs', 1100: 'rb5', 2048: 'rb4+pairs'}

# sample-images2 shape (BASELINE config 4: crop 135 x 220, map 17 x 33): default dispatch is the GENERAL matrix-core kernel
# at every batch size (small map, 33 = 32 + 1 columns); batch size -> its plan 'r<tile rows>x<column blocks>/<K slices>+v<remainder
# columns>' (K slices = waves of the tile's workgroup).  1024 is what bench.py's config-4 block launches.
EXPECTED_GEN_LAYOUT = {320: 'r2x1/4+v1', 512: 'r2x1/4+v1', 1024: 'r4x1/4+v1', 2048: 'r6x1/4+v1'}


# ------------------------------------------------------------------ CPU: the planner ----
def test_planner_invariants_every_batch_size():
    """Host logic, no GPU: for every batch size the layout covers every map row exactly once (full blocks, then pairs
    sharing their middle row), pads the template to a multiple of every wave type's rotation period, and needs one
    round of waves whenever any layout does."""
    from meterelf_amd import _hip
    (th, tw, rows, cols) = (119, 188, 250, 250)
    rh = rows - th + 1
    for n in range(1, 4301):
        d = _hip.match_layout_query(th, tw, rows, cols, n)
        assert d['kernel'] == 'mfma' and d['groups'] == (n + 31) // 32
        (rb, na, npairs, ks) = (d['rows_per_wave'], d['full_waves'], d['pair_waves'] // 2, d['k_slices'])
        assert 2 <= rb <= 5 and (npairs == 0 or rb < 5)
        assert ks in (1, 2, 4) and (ks == 1 or rb == 4)           # K slices are instantiated for the 4-row tiles
        covered = rb * na + (2 * rb + 1) * npairs
        assert rh <= covered < rh + 2 * rb + 1, (n, rb, na, npairs)
        assert d['waves'] == (na + 2 * npairs) * d['groups'] * ks and d['tiles'] == na + 2 * npairs
        # every slice's share of the (padded) template rows is a whole number of rotation periods of every wave type
        sl = d['th_pad'] // ks
        assert d['th_pad'] >= th and d['th_pad'] == sl * ks and sl % (rb + 1) == 0 and (npairs == 0 or sl % (rb + 2) == 0)
        assert d['rows_pad'] >= covered + d['th_pad'] + 1
        if 8 <= d['groups'] <= 34:      # 8 x 32 x 4 ... 34 x 30 waves: a one-round layout exists
            assert d['waves'] <= 1024, (n, d)
        if 18 <= d['groups'] <= 31:     # round 6: 576 ... 992 frames run 4-row waves on 594 ... 1023 SIMDs (power-limited chip: larger tiles, not more of them)
            assert d['layout'] == 'rb4' and d['waves'] == 33 * d['groups'], (n, d)
    for (n, want) in EXPECTED_LAYOUT.items():
        assert _hip.match_layout_query(th, tw, rows, cols, n)['layout'] == want, n


def _check_gen_plan(th, tw, rows, cols, n):
    """Invariants of the general matrix-core kernel's plan (gen_plan, k_match_gen.hip) for one shape and batch size."""
    from meterelf_amd import _hip
    (d, tasks) = _hip.match_gen_plan_query(th, tw, rows, cols, n)
    (rh, rw) = (rows - th + 1, cols - tw + 1)
    assert d['groups'] == (n + 31) // 32 and d['waves'] == len(tasks) * d['groups']
    assert d['rows_per_wave'] in (2, 4, 6, 8) and d['blocks_per_tile'] in (1, 2)
    # the slices of a tile are the waves of ONE workgroup: at most 8 (small tile shapes, two waves per SIMD) or 4
    assert 1 <= d['slices'] <= (8 if d['rows_per_wave'] * d['blocks_per_tile'] <= 4 else 4)
    assert (d['rows_per_wave'], d['blocks_per_tile']) != (8, 2)      # that shape spills: not offered
    (nd, ndv, vcols) = (d['nd'], d['v_blocks'], d['v_columns'])
    assert nd == (tw + 62) // 32 and 0 <= vcols <= 4
    vx0 = 32 * (rw // 32) if vcols else rw
    cover = np.zeros((rh, rw), np.int32)
    tiles = {}
    for t in tasks:
        tiles.setdefault(int(t['tile']), []).append(t)
    assert sorted(tiles) == list(range(d['tiles'])), 'tile ids are the slots of the (max, argmax) partials: dense'
    assert d['tiles'] < 32768                       # GenTask::tile is 16 bits wide
    for (ti, ts) in tiles.items():
        t0 = ts[0]
        ns = int(t0['nslices'])
        assert len(ts) == ns and sorted(int(t['slice']) for t in ts) == list(range(ns)), ti
        assert ns == d['slices']                      # one workgroup size per launch
        for t in ts:    # every slice of a tile describes the same tile
            assert tuple(int(t[k]) for k in ('y0', 'rows', 'rows_computed', 'xb0', 'nxb', 'lds_bytes')) == \
                   tuple(int(t0[k]) for k in ('y0', 'rows', 'rows_computed', 'xb0', 'nxb', 'lds_bytes')), ti
        (y0, R, Rc, xb0, nxb) = (int(t0[k]) for k in ('y0', 'rows', 'rows_computed', 'xb0', 'nxb'))
        assert 0 <= y0 < rh and y0 < 32768
        if R:      # H form: R map rows x nxb column blocks
            assert 1 <= R <= Rc == d['rows_per_wave'] and 1 <= nxb <= d['blocks_per_tile'] and y0 + R <= rh
            (c0, c1) = (32 * xb0, min(32 * (xb0 + nxb), vx0))
            assert c0 < c1
            cover[y0:y0 + R, c0:c1] += 1
            (klen, nq) = (nd * th, Rc * nxb * 4)
        else:      # V form: one remainder column x 32 map rows
            assert 0 <= xb0 < vcols and y0 % 32 == 0
            nrow = min(32, rh - y0)
            cover[y0:y0 + nrow, vx0 + xb0] += 1
            (klen, nq) = ((nrow + th - 1) * ndv, 4)
        # the slices partition the tile's K range, none of them empty
        spans = sorted((int(t['k_lo']), int(t['k_hi'])) for t in ts)
        assert spans[0][0] == 0 and spans[-1][1] == klen, (ti, spans, klen)
        for ((a0, a1), (b0, b1)) in zip(spans, spans[1:]):
            assert a1 == b0, (ti, spans)
        assert all(a < b for (a, b) in spans), (ti, spans)
        # the slices add up in the workgroup's LDS: room for the tile's accumulators (nq KiB) + the waves' maxima, within a CU's 160 KiB
        lds = int(t0['lds_bytes'])
        assert (lds == 0) if ns == 1 else (nq * 1024 + ns * 32 * 8 <= lds <= 160 * 1024), (ti, ns, nq, lds)
    assert (cover == 1).all(), 'every map position computed exactly once'
    # the last image row any wave requests lies inside the zero-padded L plane of the group
    assert d['rows_pad'] >= rows
    return d


def test_gen_planner_invariants_every_batch_size():
    """Host logic, no GPU (the reference has ONE code path for every batch size, meterelf/_utils.py:91-97; here
    gen_plan picks tile rows / column blocks / K slices from the number of frame groups): for every batch size 1..4300 and the
    crops of BASELINE configs 3 and 4, every map position is covered exactly once, the K slices of every tile partition its K
    range, the workgroup's LDS holds the tile, and the 16-bit tile fields hold."""
    from meterelf_amd import _hip
    seen = {}
    for (rows, cols) in ((250, 250), (135, 220)):
        last = None
        for n in range(1, 4301):
            if last is not None and (n + 31) // 32 == last[0]:
                # the plan depends on n only through the number of 32-frame groups: same groups, same plan
                (d, _t) = _hip.match_gen_plan_query(119, 188, rows, cols, n)
                assert (d['layout'], d['tiles'], d['waves']) == last[1], n
                continue
            d = _check_gen_plan(119, 188, rows, cols, n)
            last = ((n + 31) // 32, (d['layout'], d['tiles'], d['waves']))
            seen.setdefault((rows, cols), set()).add(d['layout'])
    # what BENCH config 4 launches (1024 frames of the 135 x 220 crop, 32 groups): test_general_kernel_layouts_full_path runs exactly this
    (d, _t) = _hip.match_gen_plan_query(119, 188, 135, 220, 1024)
    assert (d['default_kernel'], d['layout']) == ('gen', EXPECTED_GEN_LAYOUT[1024]), d
    for (n, want) in EXPECTED_GEN_LAYOUT.items():
        (d, _t) = _hip.match_gen_plan_query(119, 188, 135, 220, n)
        assert (d['default_kernel'], d['layout']) == ('gen', want), (n, d)
    assert len(seen[(135, 220)]) >= 4 and len(seen[(250, 250)]) >= 4    # the sweep does exercise different plans


@pytest.mark.parametrize('th,tw,rows,cols', [(9, 11, 9, 11), (9, 11, 40, 75), (40, 64, 250, 250), (119, 256, 300, 468), (60, 100, 135, 231),
                                             (119, 188, 130, 1000), (31, 33, 62, 97)], ids=lambda v: str(v))
def test_gen_planner_invariants_other_shapes(th, tw, rows, cols):
    """The same invariants for other templates and maps: 1 x 1 map, 1-4 remainder columns, several V-form row blocks, two
    column blocks per tile, templates up to 256 columns."""
    for n in (1, 32, 33, 100, 512, 1024, 3000):
        _check_gen_plan(th, tw, rows, cols, n)


def test_planner_other_shapes():
    from meterelf_amd import _hip
    # one column block (map <= 32 columns): no pairs possible
    d = _hip.match_layout_query(119, 188, 250, 210, 1024)
    assert d['kernel'] == 'mfma' and d['pair_waves'] == 0
    # outside the tuned kernel's shape class
    assert _hip.match_layout_query(119, 188, 135, 400, 64)['kernel'] == 'gen'
    assert _hip.match_layout_query(40, 64, 250, 250, 64)['kernel'] == 'gen'
    assert _hip.match_layout_query(119, 300, 250, 400, 64)['kernel'] == 'dot4'


# ------------------------------------------------------------------ GPU ----
class _DevBuf:
    def __init__(self, hip, nbytes):
        self.hip = hip
        self.p = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.p), C.c_size_t(nbytes)) == 0, 'hipMalloc of %d bytes' % nbytes

    def upload(self, arr):
        assert self.hip.hipMemcpy(self.p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0

    def free(self):
        self.hip.hipFree(self.p)


@pytest.fixture(scope='module')
def lay():
    """sample-images1 readers: default dispatch, the general matrix-core kernel forced, the VALU kernel forced; and
    256 distinct synthesised frames + two special ones."""
    from meterelf_amd import MeterReader, _hip, _params
    from oracle import pyoracle as po
    from tests.test_gpu_parity import _good, synth_frames
    if _hip.device_count() < 1:
        pytest.fail('GPU tests need an MI355X: no HIP device visible (no CPU fallback exists)')
    pfile = os.path.join(GOLDEN, 'sample-images1', 'params.yml')
    params = _params.load(pfile)
    files = sorted(glob.glob(os.path.join(GOLDEN, 'sample-images1', '*.jpg')))
    readers = {}
    saved = os.environ.pop('MELF_MATCH', None)
    try:
        readers['default'] = MeterReader(params)
        for kind in ('gen', 'dot4'):
            os.environ['MELF_MATCH'] = kind     # read when the context is created
            readers[kind] = MeterReader(params)
    finally:
        os.environ.pop('MELF_MATCH', None)
        if saved is not None:
            os.environ['MELF_MATCH'] = saved
    base = synth_frames(_good(files), 256, 4242, shift=10, sigma=3.0)
    special = np.empty((2,) + base.shape[1:], np.uint8)
    special[0] = 97                                                                   # a constant frame: every map value ties
    special[1] = np.random.default_rng(5).integers(0, 256, size=base.shape[1:], dtype=np.uint8)  # noise: below the threshold
    out = dict(readers=readers, base=base, special=special, oparams=po.Params(pfile), params=params, files=files)
    yield out
    for r in readers.values():
        r.close()


def _batch(lay, n, seed):
    """n frames: every base frame in a seeded order (repeats when n > 256), the constant frame and the noise frame at
    seeded places -- so different 32-frame groups and different lanes of a group see different content."""
    rng = np.random.default_rng(seed)
    pick = np.concatenate([rng.permutation(256) for _ in range((n + 255) // 256)])[:n]
    frames = lay['base'][pick]
    where = rng.choice(n, 4, replace=False)
    frames[where[0]] = lay['special'][0]
    frames[where[1]] = lay['special'][1]
    frames[where[2]] = lay['special'][0]
    frames[where[3]] = lay['special'][1]
    return frames, where


@pytest.mark.gpu
@pytest.mark.parametrize('n', sorted(EXPECTED_LAYOUT), ids=lambda n: 'n%d_%s' % (n, EXPECTED_LAYOUT[n]))
def test_tuned_kernel_layouts_full_path(lay, n):
    """n HBM-resident frames in ONE melf_process_batch_dev call at default dispatch: the records are byte-identical to
    those of the general matrix-core kernel and of the VALU kernel on the same device buffer, for ALL frames; the
    oracle agrees on 64 sampled frames plus the constant and the below-threshold ones."""
    from oracle import pyoracle as po
    from tests.test_gpu_parity import _compare_records
    hip = hip_runtime()
    (frames, where) = _batch(lay, n, 1000 + n)
    (H, W) = frames.shape[1:3]
    buf = _DevBuf(hip, frames.nbytes)
    try:
        buf.upload(frames)
        got = {}
        for (kind, reader) in lay['readers'].items():
            got[kind] = reader.ctx.process_batch_dev(buf.p.value, n, H, W)
            info = reader.ctx.last_match()
            assert info['n'] == n and info['groups'] == (n + 31) // 32
            if kind == 'default':
                assert (info['kernel'], info['layout']) == ('mfma', EXPECTED_LAYOUT[n]), info
            else:
                assert info['kernel'] == kind, info
    finally:
        buf.free()
    assert got['default'].tobytes() == got['gen'].tobytes(), 'tuned vs general matrix-core kernel'
    assert got['default'].tobytes() == got['dot4'].tobytes(), 'tuned vs VALU kernel'
    rng = np.random.default_rng(n)
    sample = np.unique(np.concatenate([rng.choice(n, 64, replace=False), where]))
    _compare_records(got['default'][sample], po.process_frames(frames[sample], lay['oparams']), tag='n=%d' % n)
    st = got['default']['status']
    assert (st[where] == 1).all()                      # constant and noise frames: Dials not found
    assert float(got['default']['match_val'][where[0]]) == 0.0
    assert (int(got['default']['match_x'][where[0]]), int(got['default']['match_y'][where[0]])) == (0, 0)   # first of all ties
    assert (st == 0).sum() > n * 0.8


@pytest.mark.gpu
@pytest.mark.parametrize('n', [512, 1024], ids=lambda n: 'n%d_%s' % (n, EXPECTED_LAYOUT[n]))
def test_tuned_kernel_whole_map(lay, n):
    """melf_match_ccoeff(want_map=True) at default dispatch: the whole float32 correlation map of every image bit-equal
    to the VALU kernel's, and to the oracle's on four of them."""
    from oracle import pyoracle as po
    (frames, where) = _batch(lay, n, 2000 + n)
    p = lay['params']
    ((x0, y0), (x1, y1)) = p.meter_rect.top_left, p.meter_rect.bottom_right
    # the match stage's input is the L plane of the crop: take the green channel of the crops as "L" (any u8 image does)
    imgs = np.ascontiguousarray(frames[:, y0:y1, x0:x1, 1])
    ctx = lay['readers']['default'].ctx
    (mv, mx, my, rmap) = ctx.match_ccoeff(imgs, want_map=True)
    info = ctx.last_match()
    assert (info['kernel'], info['layout']) == ('mfma', EXPECTED_LAYOUT[n]), info
    (mvd, mxd, myd, rmapd) = lay['readers']['dot4'].ctx.match_ccoeff(imgs, want_map=True)
    assert lay['readers']['dot4'].ctx.last_match()['kernel'] == 'dot4'
    assert np.array_equal(rmap.view(np.uint32), rmapd.view(np.uint32))
    assert (mv.tobytes(), mx.tobytes(), my.tobytes()) == (mvd.tobytes(), mxd.tobytes(), myd.tobytes())
    from meterelf_amd._engine import load_template
    tpl = load_template(p)
    for i in (0, int(where[0]), n // 2, n - 1):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmap[i], emap), i
        assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey), i


ALL_LAYOUTS = [(rb, pairs, cols, 1) for rb in (2, 3, 4, 5) for (pairs, cols) in ((0, 250), (3, 250), (0, 215))
               if not (pairs and rb == 5)] + [(4, pairs, cols, ks) for ks in (2, 4) for (pairs, cols) in ((0, 250), (3, 250), (0, 215))]


@pytest.mark.gpu
@pytest.mark.parametrize('rb,pairs,cols,ks', ALL_LAYOUTS, ids=lambda v: str(v))
def test_every_instantiation_forced(lay, monkeypatch, rb, pairs, cols, ks):
    """Every instantiation of the tuned kernel (RB = 2..5, with pairs, one or two column blocks; 4-row tiles in 2 and 4 K
    slices) forced at a small batch (MELF_MATCH=fast, MELF_MATCH_LAYOUT=rb,np,ks): whole maps bit-equal to the VALU kernel's,
    twice in a row (work buffers reused).  The layouts are the planner's choices at other batch sizes; forcing them keeps the
    check cheap."""
    from meterelf_amd import MeterReader
    (frames, where) = _batch(lay, 70, 3000 + rb * 10 + pairs)
    p = lay['params']
    ((x0, y0), (x1, y1)) = p.meter_rect.top_left, p.meter_rect.bottom_right
    imgs = np.ascontiguousarray(frames[:, y0:y1, x0:x0 + cols, 2])
    (mvd, mxd, myd, rmapd) = lay['readers']['dot4'].ctx.match_ccoeff(imgs, want_map=True)
    monkeypatch.setenv('MELF_MATCH', 'fast')
    monkeypatch.setenv('MELF_MATCH_LAYOUT', '%d,%d,%d' % (rb, pairs, ks))
    r = MeterReader(p)
    try:
        for rep in range(2):
            (mv, mx, my, rmap) = r.ctx.match_ccoeff(imgs, want_map=True)
            info = r.ctx.last_match()
            assert info['kernel'] == 'mfma' and info['rows_per_wave'] == rb and info['k_slices'] == ks, info
            assert (info['pair_waves'] > 0) == (pairs > 0 and cols == 250), info
            assert np.array_equal(rmap.view(np.uint32), rmapd.view(np.uint32)), (rb, pairs, cols, rep)
            assert (mv.tobytes(), mx.tobytes(), my.tobytes()) == (mvd.tobytes(), mxd.tobytes(), myd.tobytes())
    finally:
        r.close()


@pytest.mark.gpu
def test_jpeg_process_batch_default_chunk(lay):
    """melf_jpeg_process_batch with 1024 sample-images1 files -- get_meter_values' default chunk: GPU decode + the tuned
    kernel in the headline layout -- against the records of the same files decoded on the host."""
    from meterelf_amd import _hip
    from meterelf_amd._image import imread_bgr
    files = [f for f in lay['files'] if os.path.basename(f) not in REJECTED]
    blobs = [open(f, 'rb').read() for f in files]
    shapes = [_hip.jpeg_probe(b)[:3] for b in blobs]
    (H, W, _ok) = shapes[0]
    keep = [i for (i, s) in enumerate(shapes) if s == (H, W, True)]
    (files, blobs) = ([files[i] for i in keep], [blobs[i] for i in keep])
    rng = np.random.default_rng(9)
    pick = rng.integers(0, len(blobs), 1024)
    pick[:len(blobs)] = np.arange(len(blobs))
    ctx = lay['readers']['default'].ctx
    (recs, status) = ctx.jpeg_process_batch([blobs[i] for i in pick], H, W)
    info = ctx.last_match()
    assert (status == 0).all()
    assert (info['kernel'], info['layout'], info['n']) == ('mfma', 'rb4+pairs', 1024), info
    host = np.stack([imread_bgr(f) for f in files])
    href = lay['readers']['gen'].read_frames(host)       # 79 frames: the general kernel
    assert lay['readers']['gen'].ctx.last_match()['kernel'] == 'gen'
    assert recs.tobytes() == href[pick].tobytes()
    assert (recs['status'] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('rows,cols,want', [(130, 2200, 'dot4'), (200, 2000, 'gen'), (125, 4000, 'dot4')],
                         ids=lambda v: str(v))
def test_wide_crops_pick_a_kernel_that_can_launch(lay, rows, cols, want):
    """Crops wider than the prep kernel's LDS row (about 2000 px, e.g. a wide meter_rect or 4K frames): the dispatch
    must hand them to a kernel that can launch -- the VALU kernel beyond the limit -- not fail with an invalid launch.
    Whole map against the oracle."""
    from meterelf_amd._engine import load_template
    from oracle import pyoracle as po
    rng = np.random.default_rng(rows + cols)
    tpl = load_template(lay['params'])
    imgs = rng.integers(0, 256, size=(3, rows, cols), dtype=np.uint8)
    imgs[1, 4:4 + tpl.shape[0], cols - tpl.shape[1] - 7:cols - 7] = tpl      # an exact occurrence near the right edge
    ctx = lay['readers']['default'].ctx
    (mv, mx, my, rmap) = ctx.match_ccoeff(imgs, want_map=True)
    assert ctx.last_match()['kernel'] == want
    for i in range(3):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmap[i], emap), i
        assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey), i
    assert (int(mx[1]), int(my[1])) == (cols - tpl.shape[1] - 7, 4)


@pytest.mark.gpu
@pytest.mark.parametrize('th,tw,rows,cols', [(100, 170, 160, 220), (91, 190, 130, 230), (61, 163, 200, 190), (120, 188, 250, 250),
                                             (121, 180, 200, 240), (133, 175, 170, 210)],
                         ids=lambda v: str(v))
def test_other_template_heights_in_every_layout(tmp_path, monkeypatch, th, tw, rows, cols):
    """Templates whose height is NOT one short of the padded height (the fixture's 119 -> 120): up to 29 zero template
    rows -- with K slices (round 5) a whole slice of a short template can be padding, and 121 / 133 rows in four slices pad to 240 --
    for which the waves must add an all-zero row-window-sum row, in every layout (forced).  Whole maps against the oracle and the
    VALU kernel."""
    from oracle import pyoracle as po
    from tests.test_gpu_parity import _reader_with_template
    rng = np.random.default_rng(th * 1000 + tw)
    tpl = rng.integers(0, 256, size=(th, tw), dtype=np.uint8)
    n = 40
    imgs = rng.integers(0, 256, size=(n, rows, cols), dtype=np.uint8)
    imgs[3, 7:7 + th, 5:5 + tw] = tpl
    imgs[-1] = 255
    monkeypatch.setenv('MELF_MATCH', 'dot4')
    r0 = _reader_with_template(tmp_path / 'dot4', tpl)
    try:
        (mvd, mxd, myd, rmapd) = r0.ctx.match_ccoeff(imgs, want_map=True)
    finally:
        r0.close()
    for i in (0, 3, n - 1):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmapd[i], emap), i
    assert (int(mxd[3]), int(myd[3])) == (5, 7)
    monkeypatch.setenv('MELF_MATCH', 'fast')
    r = _reader_with_template(tmp_path / 'fast', tpl)
    try:
        for (rb, pairs, ks) in ((2, 0, 1), (2, 3, 1), (3, 0, 1), (3, 2, 1), (4, 0, 1), (4, 1, 1), (5, 0, 1), (4, 0, 2), (4, 1, 2), (4, 0, 4), (4, 2, 4)):
            monkeypatch.setenv('MELF_MATCH_LAYOUT', '%d,%d,%d' % (rb, pairs, ks))
            (mv, mx, my, rmap) = r.ctx.match_ccoeff(imgs, want_map=True)
            info = r.ctx.last_match()
            assert info['kernel'] == 'mfma' and info['rows_per_wave'] == rb and info['k_slices'] == ks, info
            assert info['th_pad'] % ks == 0 and info['th_pad'] >= th, info
            assert np.array_equal(rmap.view(np.uint32), rmapd.view(np.uint32)), (rb, pairs, ks)
            assert (mv.tobytes(), mx.tobytes(), my.tobytes()) == (mvd.tobytes(), mxd.tobytes(), myd.tobytes())
    finally:
        r.close()


# ------------------------------------------------------------------ GPU: BASELINE configs 4 and 5 at the sizes bench.py launches ----
@pytest.fixture(scope='module')
def lay2():
    """sample-images2 readers (BASELINE config 4: crop 135 x 220, map 17 x 33): default dispatch (the general matrix-core
    kernel at every batch size) and the VALU kernel forced; 256 distinct synthesised frames + two special ones."""
    from meterelf_amd import MeterReader, _hip, _params
    from oracle import pyoracle as po
    from tests.test_gpu_parity import _good, synth_frames
    if _hip.device_count() < 1:
        pytest.fail('GPU tests need an MI355X: no HIP device visible (no CPU fallback exists)')
    pfile = os.path.join(GOLDEN, 'sample-images2', 'params.yml')
    params = _params.load(pfile)
    files = sorted(glob.glob(os.path.join(GOLDEN, 'sample-images2', '*.jpg')))
    readers = {}
    saved = os.environ.pop('MELF_MATCH', None)
    try:
        readers['default'] = MeterReader(params)
        os.environ['MELF_MATCH'] = 'dot4'     # read when the context is created
        readers['dot4'] = MeterReader(params)
    finally:
        os.environ.pop('MELF_MATCH', None)
        if saved is not None:
            os.environ['MELF_MATCH'] = saved
    base = synth_frames(_good(files), 256, 2025, shift=6, sigma=3.0)
    special = np.empty((2,) + base.shape[1:], np.uint8)
    special[0] = 97
    special[1] = np.random.default_rng(5).integers(0, 256, size=base.shape[1:], dtype=np.uint8)
    out = dict(readers=readers, base=base, special=special, oparams=po.Params(pfile), params=params, files=files)
    yield out
    for r in readers.values():
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize('n', sorted(EXPECTED_GEN_LAYOUT), ids=lambda n: 'n%d_%s' % (n, EXPECTED_GEN_LAYOUT[n]))
def test_general_kernel_layouts_full_path(lay2, n):
    """BASELINE config 4's launch shapes (1024 = what bench.py's config-4 block times): n HBM-resident sample-images2 frames in
    ONE melf_process_batch_dev call at default dispatch -- the general matrix-core kernel in the plan gen_plan picks for that
    many frame groups (asserted) -- records byte-identical, for ALL frames, to the VALU kernel's on the same device buffer; the
    oracle on 64 sampled frames plus the constant and the below-threshold ones."""
    from oracle import pyoracle as po
    from tests.test_gpu_parity import _compare_records
    hip = hip_runtime()
    (frames, where) = _batch(lay2, n, 4000 + n)
    (H, W) = frames.shape[1:3]
    buf = _DevBuf(hip, frames.nbytes)
    try:
        buf.upload(frames)
        got = {}
        for (kind, reader) in lay2['readers'].items():
            for rep in range(2):    # twice: the arrival counters of the sliced tiles must be back at zero for the next launch
                got[kind] = reader.ctx.process_batch_dev(buf.p.value, n, H, W)
                info = reader.ctx.last_match()
                assert info['n'] == n and info['groups'] == (n + 31) // 32
                if kind == 'default':
                    assert (info['kernel'], info['layout']) == ('gen', EXPECTED_GEN_LAYOUT[n]), info
                    if rep == 0:
                        first = got[kind]
                    else:
                        assert first.tobytes() == got[kind].tobytes(), 'second launch of the same plan'
                else:
                    assert info['kernel'] == kind, info
    finally:
        buf.free()
    assert got['default'].tobytes() == got['dot4'].tobytes(), 'general matrix-core kernel vs VALU kernel'
    rng = np.random.default_rng(n)
    sample = np.unique(np.concatenate([rng.choice(n, 64, replace=False), where]))
    _compare_records(got['default'][sample], po.process_frames(frames[sample], lay2['oparams']), tag='config 4, n=%d' % n)
    st = got['default']['status']
    assert (st[where] == 1).all()
    assert float(got['default']['match_val'][where[0]]) == 0.0
    assert (int(got['default']['match_x'][where[0]]), int(got['default']['match_y'][where[0]])) == (0, 0)
    assert (st == 0).sum() > n * 0.8


@pytest.mark.gpu
def test_general_kernel_whole_map_config4(lay2):
    """melf_match_ccoeff(want_map=True) on 1024 crops of 135 x 220 at default dispatch (the plan of bench.py's config-4 launch,
    V-form column included): the whole float32 map of every image bit-equal to the VALU kernel's, and to the oracle's on four."""
    from meterelf_amd._engine import load_template
    from oracle import pyoracle as po
    n = 1024
    (frames, where) = _batch(lay2, n, 5000)
    p = lay2['params']
    ((x0, y0), (x1, y1)) = p.meter_rect.top_left, p.meter_rect.bottom_right
    imgs = np.ascontiguousarray(frames[:, y0:y1, x0:x1, 1])
    assert imgs.shape[1:] == (135, 220)
    ctx = lay2['readers']['default'].ctx
    (mv, mx, my, rmap) = ctx.match_ccoeff(imgs, want_map=True)
    info = ctx.last_match()
    assert (info['kernel'], info['layout']) == ('gen', EXPECTED_GEN_LAYOUT[n]), info
    (mvd, mxd, myd, rmapd) = lay2['readers']['dot4'].ctx.match_ccoeff(imgs, want_map=True)
    assert lay2['readers']['dot4'].ctx.last_match()['kernel'] == 'dot4'
    assert np.array_equal(rmap.view(np.uint32), rmapd.view(np.uint32))
    assert (mv.tobytes(), mx.tobytes(), my.tobytes()) == (mvd.tobytes(), mxd.tobytes(), myd.tobytes())
    tpl = load_template(p)
    for i in (0, int(where[0]), n // 2, n - 1):
        (ev, ex, ey, emap) = po.match_ccoeff(imgs[i], tpl, want_map=True)
        assert np.array_equal(rmap[i], emap), i
        assert (float(mv[i]), int(mx[i]), int(my[i])) == (ev, ex, ey), i


@pytest.mark.gpu
def test_config5_resident_full_size(lay):
    """BASELINE config 5, one GPU's share, as bench.py's config5_block launches it: 512 frames of 1920 x 1080 resident in HBM
    (6 220 800-byte frame stride, 3.2 GB), the six-dial params of bench.config5_params_dir, ONE melf_process_batch_dev call
    (tuned kernel, 4-row tiles in two K slices since round 5).  The oracle on 32 sampled frames (six positions; the reference cannot combine != 4 dials,
    meterelf/_reading.py:166), and what cannot depend on the batch: the same frames in ragged pieces and in permuted order."""
    import shutil
    from bench import config5_params_dir
    from meterelf_amd import MeterReader, _params
    from oracle import pyoracle as po
    from tests.test_gpu_parity import POS_TOL
    hip = hip_runtime()
    (n, H, W) = (512, 1080, 1920)
    d = config5_params_dir()
    reader = None
    try:
        pfile = os.path.join(d, 'params.yml')
        params = _params.load(pfile)
        op = po.Params(pfile)
        op.load_template()
        reader = MeterReader(params)
        assert len(params.dial_names) == 6
        rng = np.random.default_rng(1080)
        (src, where) = _batch(lay, n, 6000)             # 640 x 480 config-3 frames: their meter crops become the 1080p meters
        bg = rng.integers(0, 256, size=(4, H, W, 3), dtype=np.uint8)
        frames = np.empty((n, H, W, 3), np.uint8)
        for i in range(n):
            frames[i] = bg[i % 4]
        frames[:, 420:670, 1210:1460] = src[:, 160:410, 50:300]
        buf = _DevBuf(hip, frames.nbytes)
        try:
            buf.upload(frames)
            whole = reader.ctx.process_batch_dev(buf.p.value, n, H, W)
            info = reader.ctx.last_match()
            assert (info['kernel'], info['layout'], info['n']) == ('mfma', EXPECTED_LAYOUT[512], n), info
            # ragged pieces of the same device buffer: 200 (general kernel) + 33 + 279 (tuned kernel's small layouts)
            fs = H * W * 3
            cuts = (0, 200, 233, 512)
            parts = np.concatenate([reader.ctx.process_batch_dev(buf.p.value + a * fs, b - a, H, W) for (a, b) in zip(cuts, cuts[1:])])
            assert parts.tobytes() == whole.tobytes(), 'pieces'
            sample = np.unique(np.concatenate([rng.choice(n, 32, replace=False), where]))
            ores = po.process_frames(frames[sample], op)
            perm = rng.permutation(n)
            frames = frames[perm]
            buf.upload(frames)
            assert reader.ctx.process_batch_dev(buf.p.value, n, H, W).tobytes() == whole[perm].tobytes(), 'permutation'
        finally:
            buf.free()
    finally:
        if reader is not None:
            reader.close()
        shutil.rmtree(d, ignore_errors=True)      # the oracle reads the template when it is first used: only now
    for (k, i) in enumerate(sample):
        (r, o) = (whole[i], ores[k])
        assert int(r['status']) == o.status, i
        assert (int(r['match_x']), int(r['match_y']), float(r['match_val'])) == (o.match_x, o.match_y, o.match_val), i
        if o.status == 0:
            assert np.allclose(r['pos'][:6], list(o.pos)[:6], rtol=0, atol=POS_TOL), i
            assert np.allclose(r['angle'][:6], list(o.angle)[:6], rtol=0, atol=POS_TOL), i
        elif o.status == 3:
            assert int(r['unreadable_mask']) == o.unreadable_mask, i
    assert (whole['status'][where] == 1).all() and (whole['status'] == 0).sum() > n * 0.8
