"""Host-side pieces of the alignment loop (no GPU): the robust linear fit and the
argument conventions of find_linear_fit."""
import numpy as np
import pytest

from subpixal_amd.align import iter_linear_fit, find_linear_fit


def _points(n, seed=0):
    rng = np.random.default_rng(seed)
    return rng.uniform(0, 4096, (n, 2))


def test_fit_geometries_recover_known_transforms():
    xy = _points(200)
    c = np.array([2048.0, 2048.0])
    th = np.radians(0.01)
    cases = {
        'shift': (np.eye(2), np.array([0.37, -1.21])),
        'rscale': (1.0002 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]),
                   np.array([-0.5, 0.25])),
        'general': (np.array([[1.0003, 2e-5], [-4e-5, 0.9998]]), np.array([1.5, -0.75])),
    }
    for geom, (f, t) in cases.items():
        uv = (xy - c) @ f.T + c + t
        fit = iter_linear_fit(xy, uv, fitgeom=geom, center=c)
        np.testing.assert_allclose(fit['fit_matrix'], f, atol=1e-12)
        np.testing.assert_allclose(fit['offset'], t, atol=1e-9)
        assert fit['fitmask'].all() and fit['eff_nclip'] == 0
        assert np.all(fit['rms'] < 1e-9)


def test_sigma_clipping_rejects_outliers_and_weights_matter():
    xy = _points(300, 1)
    rng = np.random.default_rng(2)
    t = np.array([0.8, -0.3])
    uv = xy + t + 0.01 * rng.standard_normal(xy.shape)
    bad = rng.choice(300, 12, replace=False)
    uv[bad] += rng.uniform(3, 6, (12, 2))
    fit = iter_linear_fit(xy, uv, fitgeom='shift', nclip=3, sigma=3.0)
    assert not fit['fitmask'][bad].any()
    assert fit['fitmask'].sum() >= 280
    np.testing.assert_allclose(fit['offset'], t, atol=3e-3)
    unclipped = iter_linear_fit(xy, uv, fitgeom='shift', nclip=0)
    assert np.abs(unclipped['offset'] - t).max() > 0.05
    w = np.ones(300)
    w[bad] = 0.0
    weighted = iter_linear_fit(xy, uv, wuv=w, fitgeom='shift', nclip=0)
    np.testing.assert_allclose(weighted['offset'], t, atol=3e-3)


def test_argument_errors():
    xy = _points(5)
    with pytest.raises(ValueError):
        iter_linear_fit(xy, xy[:4])
    with pytest.raises(ValueError):
        iter_linear_fit(xy, xy, fitgeom='affine')
    with pytest.raises(ValueError):
        iter_linear_fit(xy[:2], xy[:2], fitgeom='general')
    a = np.zeros((8, 8), np.float32)
    with pytest.raises(ValueError, match="number of image cutouts"):
        find_linear_fit([a, a], [(a, a, a, a)])
    with pytest.raises(ValueError, match="four dithered blots"):
        find_linear_fit([a], [(a, a)])


def test_affine_from_map_and_shift_affine():
    from subpixal_amd import blot
    a = blot.shift_affine(3, x0=[1.0, 2.0, 3.0], y0=0.5, scale=0.8)
    assert a.shape == (3, 6)
    assert np.allclose(a[1], [0.8, 0, 2.0, 0, 0.8, 0.5])
    true = np.array([0.99, 0.02, 5.0, -0.03, 1.01, 7.0])

    def linear(x, y):
        return true[0] * x + true[1] * y + true[2], true[3] * x + true[4] * y + true[5]

    fit, res = blot.affine_from_map(linear, (64, 48))
    assert np.allclose(fit, true, atol=1e-12) and res < 1e-10

    def distorted(x, y):
        xs, ys = linear(x, y)
        return xs + 1e-5 * (x - 24) ** 2, ys

    fit2, res2 = blot.affine_from_map(distorted, (64, 48))
    assert 1e-4 < res2 < 1e-2 and np.allclose(fit2[[0, 1, 3, 4]], true[[0, 1, 3, 4]], atol=1e-3)


def test_zero_weight_points_are_not_measurements():
    """sources without a usable displacement (non-finite cutout, skipped shape) enter
    the fit with zero weight; they must neither move the fit nor count in fitmask / rms."""
    from subpixal_amd.align import usable_status, ST_SKIPPED
    xy = _points(50, 3)
    t = np.array([0.4, -0.2])
    uv = xy + t
    uv[[3, 17]] += 1e6                       # what a meaningless shift looks like
    w = np.ones(50)
    w[[3, 17]] = 0.0
    fit = iter_linear_fit(xy, uv, wuv=w, fitgeom='general', nclip=3, sigma=3.0)
    np.testing.assert_allclose(fit['offset'] + (fit['fit_matrix'] - np.eye(2)) @ xy.mean(0), t, atol=1e-6)
    assert not fit['fitmask'][[3, 17]].any() and fit['fitmask'].sum() == 48
    assert np.all(fit['rms'] < 1e-6)
    with pytest.raises(ValueError, match="non-zero weight"):
        iter_linear_fit(xy[:3], uv[:3], wuv=np.array([1.0, 0.0, 0.0]), fitgeom='general')
    st = np.array([0, 1, 2, 3, 4, 5, 6, ST_SKIPPED])
    assert list(usable_status(st)) == [True, True, True, True, False, False, False, False]


def test_host_fit_against_the_oracles_independent_restatement():
    """SURVEY 8 f-4 / VERDICT r2 item 9: `subpixal_amd.align.iter_linear_fit` against
    `oracle.iter_linear_fit`, an independent restatement (augmented lstsq, Procrustes/SVD) of the documented
    behaviour of `tweakwcs.linearfit.iter_linear_fit` (align.py:720-724).  tweakwcs itself is absent and the
    reference holds no fixture for it: parity WITH TWEAKWCS stays unpinned; this pins the host code against
    a second implementation of the same definition."""
    from oracle import subpixal_oracle as orc
    rng = np.random.default_rng(11)
    c = np.array([2048.0, 2048.0])
    th = np.radians(0.02)
    truths = {
        'shift': (np.eye(2), np.array([0.37, -1.21])),
        'rscale': (1.0003 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]), np.array([-0.5, 0.25])),
        'general': (np.array([[1.0003, 2e-5], [-4e-5, 0.9998]]), np.array([1.5, -0.75])),
    }
    for geom, (f, t) in truths.items():
        for trial in range(4):
            n = int(rng.integers(40, 400))
            xy = rng.uniform(0, 4096, (n, 2))
            uv = (xy - c) @ f.T + c + t + 0.02 * rng.standard_normal((n, 2))
            bad = rng.choice(n, n // 15, replace=False)
            uv[bad] += rng.uniform(-8, 8, (len(bad), 2))
            w = rng.uniform(0.2, 3.0, n) if trial % 2 else None
            if w is not None and trial == 3:
                w[rng.choice(n, 5, replace=False)] = 0.0          # not measurements
            got = iter_linear_fit(xy, uv, wuv=w, fitgeom=geom, center=c, nclip=3, sigma=3.0)
            ef, et, emask, eclip = orc.iter_linear_fit(xy, uv, wuv=w, fitgeom=geom, center=c, nclip=3, sigma=3.0)
            assert np.array_equal(got['fitmask'], emask), (geom, trial)
            assert got['eff_nclip'] == eclip
            np.testing.assert_allclose(got['fit_matrix'], ef, atol=1e-11, rtol=0)
            np.testing.assert_allclose(got['offset'], et, atol=1e-8, rtol=0)
            np.testing.assert_allclose(got['fit_matrix'], f, atol=2e-5)
            np.testing.assert_allclose(got['offset'], t, atol=2e-2)
