"""Half-pixel dithered blots on the GPU (spx_blot_affine4_f32): parity with
the oracle's independent float64 restatement, polynomial exactness, the dither convention of
align.py:664-676 through the reference-mode kernel, and find_linear_fit(affine=...)."""
import numpy as np
import pytest

import datagen
from oracle import subpixal_oracle as orc

pytestmark = pytest.mark.gpu


def test_blot_vs_oracle_interior_edges_outside():
    from subpixal_amd import blot
    rng = np.random.default_rng(1)
    n, sny, snx, ny, nx = 6, 40, 36, 24, 20
    src = rng.normal(size=(n, sny, snx)).astype(np.float32)
    aff = np.array([[1, 0, 6.3, 0, 1, 7.7],
                    [0.98, 0.05, -1.0, -0.04, 1.01, 2.2],
                    [1.3, 0.0, 0.2, 0.0, 1.6, 0.1],
                    [1, 0, 0.0, 0, 1, 0.0],                     # starts exactly on the corner
                    [0.7, -0.7, 20.0, 0.7, 0.7, 3.0],           # 45 degree rotation
                    [1, 0, 30.5, 0, 1, 30.5]])                  # mostly outside
    gain = rng.uniform(0.5, 2.0, n).astype(np.float32)
    got = blot.blot_affine4_batch(src, aff, (ny, nx), gain)
    exp = orc.blot_affine4(src, aff, ny, nx, gain)
    assert got.shape == (n, 4, ny, nx) and got.dtype == np.float32
    assert np.abs(got - exp).max() < 5e-6 * np.abs(exp).max()
    assert ((exp == 0) == (got == 0)).all()


def test_blot_reproduces_quintic_polynomials():
    from subpixal_amd import blot
    yy, xx = np.mgrid[0:64, 0:64].astype(float)

    def poly(x, y):
        return 1.0 + 0.05 * x - 0.02 * y + 2e-3 * x * y + 1e-4 * x ** 3 - 1e-6 * y ** 4 * x + 1e-7 * x ** 5

    a = np.array([[1.0, 0.0, 10.3, 0.0, 1.0, 12.8]])
    g = blot.blot_affine4_batch(poly(xx, yy)[None].astype(np.float32), a, (32, 32))
    for q, (ox, oy) in enumerate(((0, 0), (0.5, 0), (0, 0.5), (0.5, 0.5))):
        xt = np.arange(32)[None, :] + ox + 10.3
        yt = np.arange(32)[:, None] + oy + 12.8
        assert np.abs(g[0, q] - poly(xt, yt)).max() < 3e-6 * np.abs(poly(xt, yt)).max()


def test_blots_feed_find_displacement():
    """Drizzled-frame spots resampled onto image grids: the four GPU blots through the
    reference-mode kernel recover the known displacement, and agree with the same pipeline
    on analytically dithered images."""
    import torch
    from subpixal_amd import blot, cc
    rng = np.random.default_rng(5)
    n, size, big = 64, 48, 80
    sig = 2.5
    src = np.empty((n, big, big), np.float32)
    ref = np.empty((n, size, size), np.float32)
    ana = np.empty((n, 4, size, size), np.float32)
    aff = blot.shift_affine(n, x0=16.0, y0=16.0)               # image pixel (x, y) = drizzled (x+16, y+16)
    truth = rng.uniform(-1.5, 1.5, (n, 2))
    for k in range(n):
        cx, cy = (size - 1) / 2 + rng.uniform(-2, 2), (size - 1) / 2 + rng.uniform(-2, 2)
        src[k] = datagen.spot(big, big, cx + 16.0, cy + 16.0, sig)           # the drizzled model
        ref[k] = datagen.spot(size, size, cx + truth[k, 0], cy + truth[k, 1], sig)   # the exposure
        # blt(ox, oy)[x] = model(x + ox): a spot at c appears at c - ox (datagen.dither_set)
        for q, (ox, oy) in enumerate(((0, 0), (0.5, 0), (0, 0.5), (0.5, 0.5))):
            ana[k, q] = datagen.spot(size, size, cx - ox, cy - oy, sig)
    im4 = blot.blot_affine4_batch(torch.as_tensor(src).cuda(), aff, (size, size))
    # the quintic's own error on a sigma = 2.5 Gaussian at half-pixel offsets is ~5e-4 of the peak
    assert float((im4.cpu() - torch.as_tensor(ana)).abs().max()) < 1e-3
    d_blot = cc.find_displacement_batch(torch.as_tensor(ref).cuda(), im4, cc_type='CC').cpu().numpy()
    d_ana = cc.find_displacement_batch(ref, ana, cc_type='CC')
    assert np.abs(d_blot - d_ana).max() < 5e-3
    # the oracle on the analytic dithers defines the expected displacement (sign: cc.py:89-93)
    d_orc, _ = orc.find_displacement_batch(ref[:8], ana[:8], cc_type='CC')
    assert np.abs(d_ana[:8] - d_orc).max() < 3e-5
    assert np.abs(np.abs(d_blot) - np.abs(truth)).max() < 0.05               # 2x-interlace accuracy


def test_find_linear_fit_with_affine_blots():
    from subpixal_amd import blot
    from subpixal_amd.align import find_linear_fit
    rng = np.random.default_rng(9)
    n, size, big, sig = 40, 48, 80, 2.5
    shift = np.array([0.62, -0.41])
    imgs, drzs = [], []
    for k in range(n):
        cx, cy = (size - 1) / 2 + rng.uniform(-2, 2), (size - 1) / 2 + rng.uniform(-2, 2)
        drzs.append(datagen.spot(big, big, cx + 16.0, cy + 16.0, sig).astype(np.float32))
        imgs.append(datagen.spot(size, size, cx + shift[0], cy + shift[1], sig).astype(np.float32))
    fit, icc, blt00 = find_linear_fit(imgs, drzs, fitgeom='shift', use_weights=False, cc_type='NCC',
                                      affine=blot.shift_affine(n, 16.0, 16.0))
    assert len(icc) == n and icc[0].shape == (2 * size, 2 * size) and blt00[0].shape == (size, size)
    d = fit['subpixal_img_dxy']
    assert np.abs(d - d.mean(axis=0)).max() < 0.02                           # all sources agree
    assert np.abs(np.abs(d.mean(axis=0)) - np.abs(shift)).max() < 0.03
    with pytest.raises(ValueError):
        find_linear_fit(imgs, drzs, affine=blot.shift_affine(n), blot=lambda a, b: a)


def test_blot_poly4_distorted_map():
    """8f-2 beyond affine: cubic-distorted maps through spx_blot_poly4_f32 vs the
    oracle's float64 Lagrange restatement evaluated through the TRUE map; and the displacement the
    reference-mode kernel finds on those blots equals the one found on the oracle's blots."""
    from subpixal_amd import blot, cc
    rng = np.random.default_rng(2)
    n, sny, snx, ny, nx = 4, 60, 56, 32, 36
    yy, xx = np.mgrid[0:sny, 0:snx].astype(float)
    src = np.stack([np.exp(-((xx - 28 - k) ** 2 + (yy - 30 + k) ** 2) / (2 * 3.0 ** 2)) for k in range(n)])
    src = (src + 0.01 * rng.normal(size=src.shape)).astype(np.float32)

    def make(k):
        def mapping(x, y):
            u, v = x - 17.5, y - 15.5
            xs = 10.0 + 0.3 * k + 0.99 * x + 0.03 * y + 2e-4 * u * u - 1e-4 * u * v + 3e-6 * u ** 3
            ys = 13.5 - 0.02 * x + 1.01 * y + 1.5e-4 * v * v + 2e-6 * v ** 3 - 1e-6 * u * u * v
            return xs, ys
        return mapping
    maps = [make(k) for k in range(n)]
    coefs = np.empty((n, 2, blot.POLY_TERMS))
    for k in range(n):
        kind, (coefs[k], deg), res = blot.map_from(maps[k], (ny, nx))
        assert kind == 'poly' and deg == 3 and res < 1e-6
        assert blot.affine_from_map(maps[k], (ny, nx))[1] > 1e-3
    got = blot.blot_poly4_batch(src, coefs, (ny, nx), 3)
    exp = orc.blot_map4(src, maps, ny, nx)
    assert got.shape == (n, 4, ny, nx) and got.dtype == np.float32
    assert np.abs(got - exp).max() < 5e-6 * np.abs(exp).max()
    # a reference cutout = the un-dithered blot displaced by a known amount; displacement via the
    # GPU blots == displacement via the oracle's blots
    ref = np.stack([np.roll(exp[k, 0], (1, -2), axis=(0, 1)) for k in range(n)]).astype(np.float32)
    d_gpu = cc.find_displacement_batch(ref, got, cc_type='NCC')
    d_orc, _ = orc.find_displacement_batch(ref, exp.astype(np.float32), 'NCC')
    assert np.abs(d_gpu - d_orc).max() < 1e-4
    with pytest.raises(ValueError):
        blot.map_from(lambda x, y: (x + 5.0 + 1e-2 * np.sin(x), y + 5.0), (ny, nx), tol=1e-6)
