"""The callers either side of the hot path on a real GPU: cutout packing ->
cross-correlation -> linear fit on a synthetic frame pair (BASELINE config 5, reduced),
and find_linear_fit with an analytic blot callable through the 5-image reference path."""
import os
import sys

import numpy as np
import pytest

import datagen
from oracle import subpixal_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_synthetic_frame_alignment():
    import align_synthetic
    out = align_synthetic.run(size=2048, nsrc=150, upsample=10, quiet=True)
    # a few sources have a neighbour inside their 64x64 cutout and carry its contamination;
    # the sigma-clipped fit still has to recover the transform
    err = np.abs(out['shifts'] - out['true_shifts']).max(axis=1)
    print('median %.3g max %.3g offset err %s matrix err %.3g' % (np.median(err), err.max(), out['fit']['offset'] - out['true_offset'], np.abs(out['fit']['fit_matrix'] - out['true_matrix']).max()))
    assert np.median(err) < 5e-4
    fit = out['fit']
    assert np.abs(fit['offset'] - out['true_offset']).max() < 1e-3
    assert np.abs(fit['fit_matrix'] - out['true_matrix']).max() < 3e-6


def test_find_linear_fit_with_blot_callable():
    from subpixal_amd.align import find_linear_fit
    from subpixal_amd.cutout import Cutout

    class Scene:          # what blot_cutout does for the reference: resample the model onto a grid
        def __init__(self, x0, y0, sigma, amp):
            self.x0, self.y0, self.sigma, self.amp = x0, y0, sigma, amp

        def on_grid(self, ct):
            ny, nx = ct.height, ct.width
            # cutout pixel (i, j) sits at image coordinate (i + blc - dx)
            return self.amp * datagen.spot(ny, nx, self.x0 - ct.blc[0] + ct.dx,
                                           self.y0 - ct.blc[1] + ct.dy, self.sigma)

    rng = np.random.default_rng(8)
    frame = np.zeros((256, 256), np.float32)
    img_cutouts, drz, truth = [], [], []
    for k in range(12):
        size = (40, 48, 64)[k % 3]
        x0, y0 = rng.uniform(70, 180, 2)
        tx, ty = rng.uniform(-1.5, 1.5, 2)
        blc = (int(x0) - size // 2, int(y0) - size // 2)
        ct = Cutout(frame, None, blc=blc, trc=(blc[0] + size - 1, blc[1] + size - 1),
                    src_pos=(x0, y0), src_weight=1.0 + k)
        # the exposure sees the source displaced by (tx, ty) w.r.t. the model
        ct.data = (1.3 * datagen.spot(size, size, x0 + tx - blc[0], y0 + ty - blc[1], 3.0)).astype(np.float32)
        img_cutouts.append(ct)
        scene = Scene(x0, y0, 3.0, 0.9)
        scene.data = np.zeros((size, size), np.float32)
        scene.mask = np.zeros((size, size), bool)
        scene.src_weight = 1.0 + k
        scene.wcs = None
        drz.append(scene)
        truth.append((tx, ty))

    def blot(dz, imct):
        c = Cutout(frame, None, blc=imct.blc, trc=imct.trc)
        c.dx, c.dy = imct.dx, imct.dy
        c.data = dz.on_grid(c).astype(np.float32)
        return c

    fit, iccs, blts = find_linear_fit(img_cutouts, drz, fitgeom='shift', cc_type='NCC', blot=blot)
    d = fit['subpixal_img_dxy']
    # ground truth through the oracle's restatement of cc.find_displacement on the same blots
    for k, ct in enumerate(img_cutouts):
        b = []
        for ddx, ddy in ((0, 0), (-0.5, 0), (0, -0.5), (-0.5, -0.5)):
            c = Cutout(frame, None, blc=ct.blc, trc=ct.trc)
            c.dx, c.dy = ddx, ddy
            b.append(drz[k].on_grid(c).astype(np.float32))
        e = orc.find_displacement(ct.data, b[0], b[1], b[2], b[3], cc_type='NCC')
        assert abs(d[k, 0] - e[0]) < 2e-5 and abs(d[k, 1] - e[1]) < 2e-5
        assert iccs[k].shape == (2 * ct.height, 2 * ct.width)
    # image displaced by +t w.r.t. the model -> the fit maps image to reference by -t... sign per
    # align.py:695-699: xyref = xyim + (dx, dy), fit maps xyim -> xyref
    np.testing.assert_allclose(d, -np.array(truth), atol=5e-3)
    assert len(blts) == 12 and 'irmse' in fit and fit['fitmask'].all()
    # grid displacement restored (align.py:679)
    assert all(ct.dx == 0 and ct.dy == 0 for ct in img_cutouts)
