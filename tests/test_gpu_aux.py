import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gather_cutouts(spx_mod=None):
    import torch
    from subpixal_amd import cutout
    rng = np.random.default_rng(3)
    frame = rng.standard_normal((400, 500)).astype(np.float32)
    frame[5, 7] = np.nan
    frame[60, 80] = np.inf
    bad = np.zeros(frame.shape, bool)
    bad[100:120, 200:220] = True
    boxes = np.array([[2, 3, 64, 60], [-4, -2, 64, 64], [470, 380, 40, 30], [190, 95, 50, 50],
                      [0, 0, 5, 5]], np.int32)
    tiles = cutout.pack_cutouts(frame, boxes, (64, 64), mask=bad, fill=0.0)
    tiles = tiles.cpu().numpy()
    for b, (x0, y0, w, h) in enumerate(boxes):
        exp = np.zeros((64, 64), np.float32)
        for ty in range(h):
            for tx in range(w):
                fy, fx = y0 + ty, x0 + tx
                if 0 <= fy < 400 and 0 <= fx < 500 and not bad[fy, fx] and np.isfinite(frame[fy, fx]):
                    exp[ty, tx] = frame[fy, fx]
        np.testing.assert_array_equal(tiles[b], exp)
