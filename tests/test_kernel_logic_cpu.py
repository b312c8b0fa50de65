"""The kernel SOURCE (subpixal_amd/csrc/*.h) run on CPU threads by the logic-check
harness (tests/cpu_emu) against the oracle and the reference goldens.  This checks
index arithmetic / algorithm, not the GPU: the parity tests proper are the
`-m gpu` tests in test_gpu_parity.py, which call the real library."""
import json
import os

import numpy as np
import pytest

import datagen
import emu
from oracle import subpixal_oracle as orc


def test_pair_mode_vs_oracle():
    ref, img, truth = datagen.pair_batch(3, 2, 64)
    for up, tol in ((1, 5e-6), (2, 2e-5), (10, 1e-4)):
        got, st = emu.pair(ref, img, up)
        exp, est = orc.xcorr_refine_batch(ref, img, up)
        assert np.max(np.abs(got - exp)) < tol, (up, np.max(np.abs(got - exp)))
        assert np.array_equal(st, est)


def test_pair_mode_grid_stride_pipeline():
    """fewer workgroups than pairs: exercises the one-pair-ahead register prefetch"""
    ref, img, truth = datagen.pair_batch(4, 5, 64)
    emu.set_grid(2)
    try:
        got, st = emu.pair(ref, img, 10)
    finally:
        emu.set_grid(0)
    exp, est = orc.xcorr_refine_batch(ref, img, 10)
    assert np.max(np.abs(got - exp)) < 1e-4
    assert np.array_equal(st, est)


def test_pair_mode_float64_refine():
    """The 64 tile and its fold path with the refine stage's matrix products accumulated in float64
    (spx_kernels.h RefineF64, make_ktab_f64: what SPX_REFINE_F64 selects): same statuses, and closer to
    the float64 definition than the float32 form on the same pairs.  A wrong table order or result-row
    map (v_mfma_f64_16x16x4_f64 puts row lk + 4 r where the float32 one puts 4 lk + r) shows as whole pixels."""
    try:
        for n in (64, 80):
            ref, img, _ = datagen.pair_batch(11 + n, 3, n)
            for up in (10, 20):
                exp, est = orc.xcorr_refine_batch(ref, img, up)
                emu.set_refine64(0)
                got32, st32 = emu.pair(ref, img, up)
                emu.set_refine64(1)
                got64, st64 = emu.pair(ref, img, up)
                d32, d64 = np.abs(got32 - exp).max(), np.abs(got64 - exp).max()
                assert np.array_equal(st64, est) and np.array_equal(st32, est)
                assert d64 < 3e-5, (n, up, d64)
                assert d64 < d32, (n, up, d64, d32)
    finally:
        emu.set_refine64(-1)


def test_pair_mode_eight_wave_kernel():
    """spx_kernels8.h (round 3 A/B, SPX_PAIR64_WAVES=8): 16 mod-4 classes in half-waves, recombined into
    the four parity planes -- same results as the four-wave kernel to float32 rounding, ragged shapes,
    every cc_type, pairs following each other inside one workgroup"""
    ref, img, truth = datagen.pair_batch(3, 3, 64)
    for up, tol in ((1, 5e-6), (10, 1e-4)):
        got, st = emu.pair(ref, img, up, tile=648)
        exp, est = orc.xcorr_refine_batch(ref, img, up)
        assert np.max(np.abs(got - exp)) < tol, (up, np.max(np.abs(got - exp)))
        assert np.array_equal(st, est)
        four, _ = emu.pair(ref, img, up, tile=64)
        assert np.max(np.abs(got - four)) < tol
    rng = np.random.default_rng(3)
    emu.set_grid(2)
    try:
        for ny, nx, up, cc in ((63, 61, 10, 1), (40, 57, 2, 2), (5, 5, 1, 0), (64, 37, 30, 1), (7, 64, 10, 0)):
            r = np.empty((5, ny, nx), np.float32)
            i = np.empty_like(r)
            for k in range(5):
                tx, ty = rng.uniform(-2, 2, 2) if min(ny, nx) > 8 else rng.uniform(-0.5, 0.5, 2)
                r[k], i[k] = datagen.pair_set(ny, nx, tx, ty, rng.uniform(1.2, min(ny, nx) / 8 + 1.2), 1.3,
                                              np.float32, noise_seed=int(rng.integers(1, 1000)), noise_level=0.01)
            got, st = emu.pair(r, i, up, cc, tile=648)
            exp, est = orc.xcorr_refine_batch(r, i, upsample=up, cc_type=datagen.CC_TYPES[cc])
            assert np.max(np.abs(got - exp)) < 2e-4 and np.array_equal(st, est), (ny, nx, up, cc)
    finally:
        emu.set_grid(0)
    ref[0, 5, 5] = np.nan
    got, st = emu.pair(ref, img, 10, tile=648)
    assert st[0] == 6 and st[1] == 0 and got[0, 0] == -31.0


def test_reference_mode_five_transform_kernel():
    """spx_kernels5.h: R = FFT(ref) once, the dithers transformed two at a time, both correlations of a pair
    from one inverse -- against the oracle's cc.find_displacement (cc.py:21-95) for every cc_type, float64
    inputs, ragged shapes and the fold path; and equal (to float32 rounding) to the eight-transform kernel"""
    try:
        for n, shape, cc, dt in ((64, None, 1, np.float32), (48, None, 2, np.float32), (64, None, 0, np.float64),
                                 (77, None, 1, np.float32), (85, None, 2, np.float32), (40, (33, 40), 0, np.float32)):
            r5, m4, _ = datagen.dither_batch(3, 2, n, dtype=dt)
            if shape:
                r5 = np.ascontiguousarray(r5[:, :shape[0], :shape[1]])
                m4 = np.ascontiguousarray(m4[:, :, :shape[0], :shape[1]])
            emu.set_disp5_packed(2)
            d, st, icc = emu.disp5(r5, m4, cc)
            emu.set_disp5_packed(0)
            d0, st0, icc0 = emu.disp5(r5, m4, cc)
            e, est = orc.find_displacement_batch(r5, m4, datagen.CC_TYPES[cc])
            eicc = orc.build_icc(r5[0], *m4[0], cc_type=datagen.CC_TYPES[cc])[0]
            assert np.array_equal(st, est) and np.array_equal(st0, est)
            assert np.abs(d - e).max() < 2e-5 and np.abs(d - d0).max() < 2e-5, (n, cc, np.abs(d - e).max())
            assert np.abs(icc[0] - eicc).max() < 3e-6 * np.abs(eicc).max()
    finally:
        emu.set_disp5_packed(1)
    bad = r5.copy()
    bad[0, 3, 3] = np.nan
    emu.set_disp5_packed(2)
    try:
        d, st, _ = emu.disp5(bad, m4, 0)
    finally:
        emu.set_disp5_packed(1)
    assert st[0] == 6 and st[1] == 0


def test_item_walk_is_a_bijection_and_keeps_runs_of_eight_on_one_l2():
    """first_item (spx_kernels.h): workgroup -> first item of its grid-stride walk.  Every grid size
    must visit each item exactly once; launches of a multiple of 64 workgroups hand workgroups
    b, b+8, ... (one XCD's, by the observed round-robin placement) runs of 8 consecutive items."""
    for nwg in (1, 7, 63, 64, 65, 100, 128, 192, 8192):
        seen = sorted(emu.first_item(b, nwg) for b in range(nwg))
        assert seen == list(range(nwg)), nwg
    for nwg in (64, 8192):
        for x in range(8):
            items = [emu.first_item(b, nwg) for b in range(x, nwg, 8)]
            assert all(items[k + 1] == items[k] + 1 for k in range(len(items) - 1) if (k + 1) % 8), (nwg, x)
            assert all(i // 8 % 8 == x for i in items)
    # a whole launch through the remapped walk gives the per-item results of the linear one
    ref, img, truth = datagen.pair_batch(9, 70, 40)
    emu.set_grid(3)
    try:
        lin, st_lin = emu.pair(ref, img, 1)
        emu.set_grid(64)
        rem, st_rem = emu.pair(ref, img, 1)
    finally:
        emu.set_grid(0)
    assert np.array_equal(lin, rem) and np.array_equal(st_lin, st_rem)


def test_pair_mode_period_192():
    """cutouts of 86..128 px: period-192 path (9 classes, radix-3 fold and combine, workspace)"""
    ref, img, truth = datagen.pair_batch(5, 1, 128)
    got, st = emu.pair(ref, img, 2)
    exp, est = orc.xcorr_refine_batch(ref, img, 2)
    assert np.max(np.abs(got - exp)) < 2e-5 and np.array_equal(st, est)
    r, i = datagen.pair_set(100, 90, 0.7, -1.1, 5.0, 1.3, np.float32)
    got, st = emu.pair(r[None], i[None], 1, 2)
    e = orc.xcorr_refine(r, i, 1, 'ZNCC')
    assert np.max(np.abs(got[0] - np.array(e))) < 2e-5
    ref, img, truth = datagen.pair_batch(6, 2, 96)
    got, st = emu.pair(ref, img, 10)
    exp, est = orc.xcorr_refine_batch(ref, img, 10)
    assert np.max(np.abs(got - exp)) < 1e-4 and np.array_equal(st, est)
    assert np.max(np.abs(got - truth)) < 2e-4


def test_pair_mode_fold_path():
    """cutouts of 65..85 px on the 64 tile: period 128 = the smallest alias-free period for the
    'same' window; the samples beyond index 63 are folded into the four parity classes in LDS"""
    rng = np.random.default_rng(8)
    for (ny, nx) in ((65, 65), (85, 85), (70, 81), (85, 9), (12, 66)):
        r, i = datagen.pair_set(ny, nx, rng.uniform(-2, 2), rng.uniform(-2, 2), min(ny, nx) / 12 + 1, 1.3,
                                np.float32, noise_seed=7, noise_level=0.01)
        for up, name, code in ((1, 'CC', 0), (2, 'NCC', 1), (10, 'ZNCC', 2)):
            got, st = emu.pair(r[None], i[None], up, code)
            s2 = []
            e = orc.xcorr_refine(r, i, up, name, _status=s2)
            assert np.max(np.abs(got[0] - np.array(e))) < (1e-4 if up > 2 else 2e-5), (ny, nx, up)
            assert st[0] == s2[-1]
    # at integer lags every alias-free period gives the same numbers: period 128 == period 192
    r, i = datagen.pair_set(80, 77, 1.3, -0.4, 5.0, 1.0, np.float32)
    a, _ = emu.pair(r[None], i[None], 1, 0)
    b, _ = emu.pair(r[None], i[None], 1, 0, tile=192)
    assert np.max(np.abs(a - b)) < 2e-6
    assert orc.fft_period(65, 85) == 128 and orc.fft_period(86, 5) == 192 and orc.fft_period(128, 128) == 192
    assert orc.fft_period(64, 64) == 128 and orc.fft_period(32, 20) == 64
    # reference mode on the fold path
    r5, m4, _ = datagen.dither_batch(3, 1, 77)
    d, st, icc = emu.disp5(r5, m4, 1)
    e, est = orc.find_displacement_batch(r5, m4, 'NCC')
    eicc = orc.build_icc(r5[0], *m4[0], cc_type='NCC')[0]
    assert np.abs(d - e).max() < 2e-5 and np.array_equal(st, est)
    assert np.abs(icc[0] - eicc).max() < 3e-6 * np.abs(eicc).max()


def test_nonfinite_pixels_are_flagged_not_fatal():
    """one NaN pixel made the 96/128 tiles index far outside their workspace.  A NaN
    makes every lag NaN; numpy.argmax then returns 0 and find_peak the integer position (0, 0)
    (centroid.py:114, 171-172): same result here, with status ST_NONFINITE, on every tile."""
    for n in (20, 64, 80, 128):
        r, i = datagen.pair_set(n, n, 1.3, -0.7, 3.0, 1.0, np.float32)
        for bad in (np.nan, np.inf):
            i2 = i.copy()
            i2[3, 4] = bad
            for up in (1, 10):
                got, st = emu.pair(np.stack([r, r]), np.stack([i2, i]), up, 0)
                s2 = []
                e = orc.xcorr_refine(r, i2, up, 'CC', _status=s2)
                assert st[0] == 6 == s2[-1] and st[1] == 0
                assert tuple(got[0]) == e == (-((n - 1) // 2), -((n - 1) // 2))
        r5, m4, _ = datagen.dither_batch(3, 1, n)
        m4 = m4.copy()
        m4[0, 2, 5, 5] = np.nan
        d, st, _ = emu.disp5(r5, m4, 1)
        assert st[0] == 6 and tuple(d[0]) == (-((n - 1) // 2), -((n - 1) // 2))
        # plain CC: only the dither holding the NaN has a NaN correlation; numpy.argmax returns the
        # first NaN of the interlaced image (row 1, column 0 for dither 01)
        d, st, _ = emu.disp5(r5, m4, 0)
        e, est = orc.find_displacement_batch(r5, m4, 'CC')
        assert st[0] == 6 == est[0] and np.array_equal(d, e)


def test_float64_inputs_use_float64_masks_and_statistics(golden_dir):
    """float64 cutouts: cc.py:135's `!= 0` mask and the pooled statistics are taken from the float64
    values (tails that underflow in float32 stay in the mask), closing the 5e-3 px ZNCC gap of r1"""
    g = np.load(os.path.join(golden_dir, 'find_displacement.npz'))
    sel = [i for i in range(len(g['dx'])) if g['dtype'][i] == 1 and g['cc_type'][i] == 2][::6]
    assert len(sel) >= 8
    for i in sel:
        ny, nx = int(g['ny'][i]), int(g['nx'][i])
        ims = datagen.dither_set(ny, nx, g['tx'][i], g['ty'][i], g['sigma'][i], g['amp'][i], np.float64,
                                 int(g['noise_seed'][i]), g['noise_level'][i], int(g['zero_mode'][i]))
        out, st, icc = emu.disp5(ims[0][None], np.stack(ims[1:])[None], 2)
        assert abs(out[0, 0] - g['dx'][i]) < 2e-5 and abs(out[0, 1] - g['dy'][i]) < 2e-5, (ny, nx)
    r, i = datagen.pair_set(64, 64, 0.6, -1.2, 4.0, 1.0, np.float64)
    got, st = emu.pair(r[None], i[None], 10, 2)
    e = orc.xcorr_refine(r, i, 10, 'ZNCC')
    assert np.max(np.abs(got[0] - np.array(e))) < 1e-4


def test_pair_mode_32_tile():
    """cutouts up to 32 px: period-64 path, one wave per pair, 4 pairs per workgroup"""
    ref, img, truth = datagen.pair_batch(9, 6, 32)
    emu.set_grid(1)                  # 6 pairs on one workgroup: waves loop over pairs
    try:
        got, st = emu.pair(ref, img, 10)
    finally:
        emu.set_grid(0)
    exp, est = orc.xcorr_refine_batch(ref, img, 10)
    assert np.max(np.abs(got - exp)) < 1e-4 and np.array_equal(st, est)


def test_pair_mode_shapes_and_cc_types():
    rng = np.random.default_rng(1)
    for (ny, nx) in ((20, 31), (64, 40), (5, 6)):
        for cc, name in ((0, 'CC'), (1, 'NCC'), (2, 'ZNCC')):
            s = min(ny, nx)
            r, i = datagen.pair_set(ny, nx, rng.uniform(-1, 1), rng.uniform(-1, 1),
                                    max(0.8, s / 12), 1.3, np.float32)
            i = i.copy()
            i[np.abs(i) < 1e-3] = 0
            got, st = emu.pair(r[None], i[None], 3, cc)
            st2 = []
            e = orc.xcorr_refine(r, i, 3, name, _status=st2)
            assert np.max(np.abs(got[0] - np.array(e))) < 1e-4
            assert st[0] == st2[-1]


def test_disp5_vs_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, 'find_displacement.npz'))
    sel = [i for i in range(len(g['dx']))
           if g['dtype'][i] == 0 and max(g['ny'][i], g['nx'][i]) <= 64][::17]
    assert len(sel) >= 8
    for i in sel:
        ny, nx = int(g['ny'][i]), int(g['nx'][i])
        ims = datagen.dither_set(ny, nx, g['tx'][i], g['ty'][i], g['sigma'][i], g['amp'][i],
                                 np.float32, int(g['noise_seed'][i]), g['noise_level'][i],
                                 int(g['zero_mode'][i]))
        out, st, icc = emu.disp5(ims[0][None], np.stack(ims[1:])[None], int(g['cc_type'][i]))
        assert abs(out[0, 0] - g['dx'][i]) < 2e-5 and abs(out[0, 1] - g['dy'][i]) < 2e-5
        assert int(np.argmax(icc[0])) == int(g['icc_argmax'][i])
        assert abs(float(icc[0].max()) - g['icc_max'][i]) <= 2e-6 * abs(g['icc_max'][i])


def test_find_peak_kernel_vs_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, 'find_peak.npz'))
    meta = json.loads(str(g['meta_json']))
    picked = list(range(3, len(meta['cases']), 61)) + [451, 458, 497, 500, 505, 517]
    for k in sorted(set(picked)):
        case = meta['cases'][k]
        if case['degenerate']:      # sign of det decided by lstsq rounding noise
            continue
        kw = case['kwargs']
        img = g['img_%03d' % k]
        ny, nx = img.shape
        fit = kw.get('peak_fit_box', 5)
        fit = tuple(fit) if isinstance(fit, list) else (fit, fit)
        sb = kw.get('peak_search_box', None)
        if sb == 'fitbox':
            sb = fit
        elif sb == 'off':
            sb = None
        elif sb == 'all':
            sb = (ny, nx)
        if sb is None:
            sb = (0, 0)
        elif isinstance(sb, list):
            sb = tuple(sb)
        elif isinstance(sb, int):
            sb = (sb, sb)
        guess = None
        if 'xmax' in kw:
            guess = np.array([[kw['xmax'], kw['ymax']]])
        else:
            sb = (0, 0)
        mask = g['mask_%03d' % k][None] if case['has_mask'] else None
        out, st = emu.find_peak(img[None], guess, fit, sb, mask)
        exp = case['expected']
        assert abs(out[0, 0] - exp[0]) < 1e-7 and abs(out[0, 1] - exp[1]) < 1e-7, (k, kw, out, exp)


def test_gather_cutouts_logic():
    rng = np.random.default_rng(3)
    frame = rng.standard_normal((40, 50)).astype(np.float32)
    frame[5, 7] = np.nan
    frame[6, 8] = np.inf
    fmask = np.zeros(frame.shape, np.uint8)
    fmask[10:12, 20:22] = 1
    boxes = np.array([[2, 3, 16, 12], [-4, -2, 16, 16], [44, 35, 10, 9], [18, 8, 7, 7]], np.int32)
    tiles = emu.gather(frame, fmask, boxes, 16, 16, -1.0)
    for b, (x0, y0, w, h) in enumerate(boxes):
        exp = np.zeros((16, 16), np.float32)
        for ty in range(h):
            for tx in range(w):
                fy, fx = y0 + ty, x0 + tx
                v = -1.0
                if 0 <= fy < 40 and 0 <= fx < 50 and not fmask[fy, fx] and np.isfinite(frame[fy, fx]):
                    v = frame[fy, fx]
                exp[ty, tx] = v
        np.testing.assert_array_equal(tiles[b], exp)


def test_generator_logic():
    from subpixal_amd.synth import pair_params
    ref, img, truth = emu.gen_pairs(123, 5, 3, 32, 3.0, 4.0, 3.0)
    tx, ty, sg, am = pair_params(123, 5, 3, 3.0, 4.0, 3.0)
    np.testing.assert_allclose(truth, np.stack([tx, ty], 1), atol=1e-15)
    for k in range(3):
        r, i = datagen.pair_set(32, 32, tx[k], ty[k], sg[k], am[k], np.float64)
        assert np.max(np.abs(ref[k] - r)) < 1e-5 and np.max(np.abs(img[k] - i)) < 1e-5


def _label_image(rng, ny, nx, nsrc):
    seg = np.zeros((ny, nx), np.int32)
    for k in range(1, nsrc + 1):
        cy, cx = rng.integers(0, ny), rng.integers(0, nx)
        h, w = rng.integers(1, 9), rng.integers(1, 12)
        seg[max(0, cy - h):cy + h, max(0, cx - w):cx + w] = k        # later labels overwrite
    return seg


def test_label_bboxes_logic():
    rng = np.random.default_rng(11)
    seg = _label_image(rng, 61, 83, 40)
    boxes, counts = emu.label_bboxes(seg, 45)
    for l in range(1, 46):
        yy, xx = np.where(seg == l)
        if len(yy) == 0:
            assert counts[l] == 0 and tuple(boxes[l]) == (2**31 - 1, 2**31 - 1, -1, -1)
        else:
            assert counts[l] == len(yy)
            assert tuple(boxes[l]) == (xx.min(), yy.min(), xx.max(), yy.max())
    # segment-aware gather: pixels of other labels are filled
    frame = rng.standard_normal(seg.shape).astype(np.float32)
    ids, bx = orc.primary_boxes(seg, pad=1)
    assert len(ids) > 5
    tiles = emu.gather(frame, None, bx, 24, 26, 0.0, seg, ids)
    for b, (x0, y0, w, h) in enumerate(bx):
        exp = np.zeros((24, 26), np.float32)
        sub = seg[y0:y0 + h, x0:x0 + w] == ids[b]
        exp[:h, :w] = np.where(sub, frame[y0:y0 + h, x0:x0 + w], 0.0)
        np.testing.assert_array_equal(tiles[b], exp)


def test_blot_affine4_logic_vs_oracle():
    """Dithered blots (8f-2): the kernel's Everett-form quintic against the oracle's
    independent Lagrange form in float64; interior, edge continuation, points outside."""
    rng = np.random.default_rng(0)
    n, sny, snx, ny, nx = 3, 24, 20, 12, 14
    src = rng.normal(size=(n, sny, snx)).astype(np.float32)
    aff = np.array([[1, 0, 3.3, 0, 1, 4.7],                    # interior only
                    [0.98, 0.05, -1.0, -0.04, 1.01, 2.2],      # rotation, partly outside
                    [1.3, 0.0, 0.2, 0.0, 1.6, 0.1]])           # scale, touches the far edges
    gain = np.array([1.0, 2.5, 0.5], np.float32)
    got = emu.blot_affine4(src, aff, ny, nx, gain)
    exp = orc.blot_affine4(src, aff, ny, nx, gain)
    assert got.shape == (n, 4, ny, nx)
    assert np.abs(got - exp).max() < 5e-6 * np.abs(exp).max()
    assert ((exp == 0) == (got == 0)).all()                    # same points fall outside
    # a degree-5 polynomial is reproduced exactly away from the edges
    yy, xx = np.mgrid[0:sny, 0:snx].astype(float)

    def poly(x, y):
        return 0.3 + 0.1 * x - 0.02 * y + 0.01 * x * y + 1e-3 * x ** 3 - 2e-4 * y ** 4 + 1e-5 * x ** 5

    a = np.array([[1, 0, 4.25, 0, 1, 5.5]])
    g = emu.blot_affine4(poly(xx, yy)[None].astype(np.float32), a, 8, 8)
    for q, (ox, oy) in enumerate(((0, 0), (0.5, 0), (0, 0.5), (0.5, 0.5))):
        xt = np.arange(8)[None, :] + ox + 4.25
        yt = np.arange(8)[:, None] + oy + 5.5
        assert np.abs(g[0, q] - poly(xt, yt)).max() < 2e-6 * np.abs(poly(xt, yt)).max()


def test_blot_poly4_distorted_map_vs_oracle():
    """8f-2 beyond affine: a cubic-distorted target -> source map, fitted by
    blot.poly_from_map and resampled by the polynomial-map kernel, against the oracle's float64
    Lagrange restatement evaluated through the TRUE map; the affine form of the same map is off
    by more than 1e-3 px, which is why the polynomial form exists."""
    from subpixal_amd import blot
    rng = np.random.default_rng(2)
    n, sny, snx, ny, nx = 3, 44, 40, 20, 24
    src = rng.normal(size=(n, sny, snx)).astype(np.float32)

    def make(k):
        def mapping(x, y):          # rotation + scale + quadratic/cubic distortion (FLT-like)
            u, v = x - 11.0, y - 9.0
            xs = 8.0 + k + 0.99 * x + 0.03 * y + 2e-4 * u * u - 1e-4 * u * v + 3e-6 * u ** 3
            ys = 9.5 - 0.02 * x + 1.01 * y + 1.5e-4 * v * v + 2e-6 * v ** 3 - 1e-6 * u * u * v
            return xs, ys
        return mapping
    maps = [make(k) for k in range(n)]
    coefs = np.empty((n, 2, blot.POLY_TERMS))
    for k in range(n):
        a, res1 = blot.affine_from_map(maps[k], (ny, nx))
        assert res1 > 1e-3                                   # not affine at the 1e-3 px level
        coefs[k], res3 = blot.poly_from_map(maps[k], (ny, nx), degree=3)
        assert res3 < 1e-9
        kind, (c, deg), res = blot.map_from(maps[k], (ny, nx))
        assert kind == 'poly' and deg == 3 and res < 1e-3     # degree 2 is not enough
    gain = rng.uniform(0.5, 2.0, n).astype(np.float32)
    got = emu.blot_poly4(src, coefs, 3, ny, nx, gain)
    exp = orc.blot_map4(src, maps, ny, nx, gain)
    assert np.abs(got - exp).max() < 5e-6 * np.abs(exp).max()
    assert ((exp == 0) == (got == 0)).all()
    # degree 1 of the polynomial form == the affine kernel
    a = np.array([[0.98, 0.05, 6.0, -0.04, 1.01, 7.2]] * n)
    c1 = np.zeros((n, 2, blot.POLY_TERMS))
    xc, yc = 0.5 * (nx - 1), 0.5 * (ny - 1)
    c1[:, 0, 0] = a[:, 2] + a[:, 0] * xc + a[:, 1] * yc
    c1[:, 0, 1], c1[:, 0, 2] = a[:, 0], a[:, 1]
    c1[:, 1, 0] = a[:, 5] + a[:, 3] * xc + a[:, 4] * yc
    c1[:, 1, 1], c1[:, 1, 2] = a[:, 3], a[:, 4]
    assert np.array_equal(emu.blot_poly4(src, c1, 1, ny, nx), emu.blot_affine4(src, a, ny, nx))
    with pytest.raises(ValueError):
        blot.map_from(lambda x, y: (x + 5.0 + 1e-2 * np.sin(x), y + 5.0), (ny, nx), tol=1e-6)


def test_general_path_above_128_px():
    """Cutouts above 128 px (the reference's cutouts have no size limit, cutout.py:159-175): FFT period
    64 C with the class count C at run time (here 4, 5 and 7: even and odd), pair and reference mode"""
    rng = np.random.default_rng(1)
    for (ny, nx) in ((129, 129), (200, 131), (30, 260)):
        r, i = datagen.pair_set(ny, nx, rng.uniform(-2, 2), rng.uniform(-2, 2), min(ny, nx) / 14 + 1, 1.0,
                                np.float32, noise_seed=3, noise_level=0.01)
        assert orc.fft_period(ny, nx) == {129: 256, 200: 320, 260: 448}[max(ny, nx)]
        for up, name, code in ((1, 'CC', 0), (10, 'ZNCC', 2)):
            got, st = emu.pair(r[None], i[None], up, code)
            s2 = []
            e = orc.xcorr_refine(r, i, up, name, _status=s2)
            assert np.max(np.abs(got[0] - np.array(e))) < (1e-4 if up > 1 else 2e-5), (ny, nx, up)
            assert st[0] == s2[-1]
    r5, m4, _ = datagen.dither_batch(3, 1, 150)
    d, st, icc = emu.disp5(r5, m4, 1)
    e, est = orc.find_displacement_batch(r5, m4, 'NCC')
    assert np.abs(d - e).max() < 2e-5 and np.array_equal(st, est)
    eicc = orc.build_icc(r5[0], *m4[0], cc_type='NCC')[0]
    assert np.abs(icc[0] - eicc).max() < 3e-6 * np.abs(eicc).max()


def test_peak_beyond_the_last_coarse_sample_is_bracketed():
    """the SPX_ST_WINDOW branch was untested.  The one way to reach it was a peak in the
    strip of the fine image that extends (U-1)/U of a pixel beyond the last coarse sample (shift
    n//2 + 0.8 px): the window could not move past the last sample and gave up after four tries with
    an integer peak, where the oracle (full fine grid) fits normally.  The window centre may now sit
    one past the last sample; kernel == oracle there, status 0, on every kernel family."""
    for n, x0 in ((64, 12.0), (32, 6.0), (80, 14.0), (100, 20.0)):
        ref = datagen.spot(n, n, x0, (n - 1) / 2, 2.5).astype(np.float32)
        img = datagen.spot(n, n, x0 + n // 2 + 0.8, (n - 1) / 2 + 0.3, 2.5).astype(np.float32)
        for r, i in ((ref, img), (ref.T.copy(), img.T.copy())):
            got, st = emu.pair(r[None], i[None], 10, 0)
            s2 = []
            e = orc.xcorr_refine(r, i, 10, 'CC', _status=s2, full_grid=True)
            assert st[0] == s2[-1] == 0, (n, st, s2)
            assert np.max(np.abs(got[0] - np.array(e))) < 1e-4, (n, got, e)
            assert max(got[0]) > n // 2 + 0.7


def test_variable_shape_batches():
    """Cutouts of different shapes in one launch per kernel family (the reference's cutouts are bounding
    boxes + padding, one shape per source: cutout.py:159-175): same numbers as cutout-by-cutout calls;
    an item the launched family cannot take is refused with SPX_ST_SHAPE, not computed wrongly."""
    rng = np.random.default_rng(2)
    for fam, shapes in ((32, [(20, 31), (5, 9), (3, 40)]), (64, [(64, 40), (33, 50), (20, 20)]),
                        (85, [(70, 80), (40, 66), (90, 20)]), (128, [(100, 90), (86, 30), (50, 50)])):
        refs, ims = [], []
        for (ny, nx) in shapes:
            t = datagen.dither_set(ny, nx, rng.uniform(-1, 1), rng.uniform(-1, 1), max(1.0, min(ny, nx) / 10), 1.0,
                                   np.float32, noise_seed=5, noise_level=0.01)
            refs.append(t[0])
            ims.append(np.stack(t[1:]))
        out, st, iccs = emu.disp5_var(refs, ims, fam, 1)
        for k, (ny, nx) in enumerate(shapes):
            if max(ny, nx) > fam:
                assert st[k] == 7 and np.all(np.isnan(out[k]))
                continue
            s2 = []
            e = orc.find_displacement(refs[k], *ims[k], cc_type='NCC', _status=s2)
            assert np.abs(out[k] - np.array(e)).max() < 2e-5 and st[k] == s2[-1], (fam, ny, nx)
            one, st1, icc1 = emu.disp5(refs[k][None], ims[k][None], 1)
            if max(ny, nx) > (0 if fam == 32 else 32 if fam == 64 else 64 if fam == 85 else 85):
                # same kernel family as the uniform call would pick: bit-identical
                assert np.array_equal(one[0], out[k]) and np.array_equal(icc1[0], iccs[k])


def test_cutouts_narrower_than_a_load_chunk():
    """Reference mode takes cutouts down to 3 px per side.  The branch-free staging loads 4-pixel chunks
    and, for a chunk that straddles the row end, the row's LAST four pixels: a 3-pixel row has none, and the
    first version read in front of the buffer (a GPU memory fault in tools/sweep_disp5.py; reproduced as a
    heap-buffer-overflow under the harness's address sanitizer).  Rows shorter than a chunk are now read
    pixel by pixel; results against the oracle on every family."""
    rng = np.random.default_rng(3)
    for fam, shapes in ((32, [(3, 3), (30, 3), (3, 31)]), (64, [(50, 3), (3, 50), (64, 4)]),
                        (85, [(80, 3), (3, 85)]), (128, [(100, 3), (3, 128), (128, 4)])):
        refs, ims = [], []
        for (ny, nx) in shapes:
            t = datagen.dither_set(ny, nx, rng.uniform(-.5, .5), rng.uniform(-.5, .5), max(0.8, min(ny, nx) / 6), 1.0,
                                   np.float32, noise_seed=5, noise_level=0.01)
            refs.append(t[0])
            ims.append(np.stack(t[1:]))
        out, st, iccs = emu.disp5_var(refs, ims, fam, 2)
        for k in range(len(shapes)):
            s2 = []
            e = orc.find_displacement(refs[k], *ims[k], cc_type='ZNCC', _status=s2)
            eicc = orc.build_icc(refs[k], *ims[k], cc_type='ZNCC')[0]
            assert np.abs(out[k] - np.array(e)).max() < 2e-5 and st[k] == s2[-1], (fam, shapes[k])
            assert np.abs(iccs[k] - eicc).max() < 3e-6 * np.abs(eicc).max()
    t = datagen.dither_set(130, 3, 0.2, -0.3, 0.9, 1.0, np.float32)           # general path
    d, st, _ = emu.disp5(t[0][None], np.stack(t[1:])[None], 1)
    assert np.abs(d[0] - np.array(orc.find_displacement(*t, cc_type='NCC'))).max() < 2e-5


def test_catalog_gather_and_blots_equal_the_fixed_shape_kernels():
    """round 3: gather_cutouts_var_kernel / blot4_var_kernel (packed variable-shape catalog) against the
    fixed-shape kernels they generalise, item by item, bit for bit; boxes overhanging the frame, masks, segments,
    affine and polynomial maps, gains, a source too small to resample"""
    rng = np.random.default_rng(12)
    frame = rng.standard_normal((90, 110)).astype(np.float32)
    frame[20, 30] = np.inf
    fmask = rng.random((90, 110)) < 0.02
    seg = (rng.random((90, 110)) < 0.5).astype(np.int32) * 3
    boxes = np.array([[10, 12, 33, 21], [-4, 70, 20, 30], [60, 5, 7, 9], [100, 80, 15, 14], [40, 40, 5, 64]], np.int32)
    ids = np.array([3, 3, 0, 3, 3], np.int32)
    packed, offs = emu.gather_var(frame, fmask, boxes, 0.0, seg, ids)
    for k, (x0, y0, w, h) in enumerate(boxes):
        want = emu.gather(frame, fmask, boxes[k:k + 1], int(h), int(w), 0.0, seg, ids[k:k + 1])[0]
        assert np.array_equal(packed[offs[k]:offs[k] + w * h].reshape(h, w), want), k
    raw, _ = emu.gather_var(frame, None, boxes, np.nan)
    assert np.isnan(raw[offs[1]]) and not np.any(raw == -7.0)
    # blots: sources = the packed cutouts above, targets of other shapes
    src_shapes = boxes[:, [3, 2]]
    dst_shapes = np.array([[20, 25], [12, 31], [8, 8], [9, 10], [40, 6]], np.int32)
    aff = np.stack([np.array([1.01, 0.02, 2.3, -0.01, 0.98, 1.7]) for _ in range(5)])
    aff[:, 2] += np.arange(5) * 0.37
    gain = np.array([1.0, 0.5, 2.0, 1.5, 3.0], np.float32)
    src = np.where(np.isfinite(packed), packed, 0.0).astype(np.float32)
    im4, doffs = emu.blot4_var(src, offs, src_shapes, aff, 0, dst_shapes, gain)
    coef = np.zeros((5, 2, 21))
    coef[:, 0, 0], coef[:, 1, 0] = 3.0, 2.5
    coef[:, 0, 1], coef[:, 1, 2] = 0.97, 1.02
    coef[:, 0, 4] = 1e-3
    im4p, _ = emu.blot4_var(src, offs, src_shapes, coef, 3, dst_shapes)
    for k in range(5):
        h, w = (int(v) for v in src_shapes[k])
        ny, nx = (int(v) for v in dst_shapes[k])
        tile = src[offs[k]:offs[k] + h * w].reshape(1, h, w)
        got = im4[4 * doffs[k]:4 * doffs[k] + 4 * ny * nx].reshape(4, ny, nx)
        gotp = im4p[4 * doffs[k]:4 * doffs[k] + 4 * ny * nx].reshape(4, ny, nx)
        if min(h, w) < 6:                          # too small for the quintic: no signal
            assert not got.any() and not gotp.any()
            continue
        assert np.array_equal(got, emu.blot_affine4(tile, aff[k:k + 1], ny, nx, gain[k:k + 1])[0]), k
        assert np.array_equal(gotp, emu.blot_poly4(tile, coef[k:k + 1], 3, ny, nx)[0]), k
