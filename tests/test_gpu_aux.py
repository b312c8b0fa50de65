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


def test_label_bboxes_and_primary_boxes():
    import torch
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_kernel_logic_cpu import _label_image
    from oracle import subpixal_oracle as orc
    from subpixal_amd import cutout
    rng = np.random.default_rng(12)
    seg = _label_image(rng, 700, 901, 900)
    ids, boxes = cutout.primary_cutout_boxes(seg, pad=2)
    eids, eboxes = orc.primary_boxes(seg, pad=2)
    np.testing.assert_array_equal(ids, eids)
    np.testing.assert_array_equal(boxes, eboxes)
    some = eids[::3]
    ids2, boxes2 = cutout.primary_cutout_boxes(seg, ids=some, pad=0.3)
    eids2, eboxes2 = orc.primary_boxes(seg, ids=some, pad=0.3)
    np.testing.assert_array_equal(ids2, eids2)
    np.testing.assert_array_equal(boxes2, eboxes2)
    frame = rng.standard_normal(seg.shape).astype(np.float32)
    tiles = cutout.pack_cutouts(frame, boxes, (32, 32), segmentation_image=seg, ids=ids).cpu().numpy()
    for b in range(0, len(ids), 17):
        x0, y0, w, h = boxes[b]
        exp = np.zeros((32, 32), np.float32)
        for ty in range(h):
            for tx in range(w):
                fy, fx = y0 + ty, x0 + tx           # padded boxes may overhang the frame
                if 0 <= fy < seg.shape[0] and 0 <= fx < seg.shape[1] and seg[fy, fx] == ids[b]:
                    exp[ty, tx] = frame[fy, fx]
        np.testing.assert_array_equal(tiles[b], exp)
    # full-size property: 4k x 4k frame, counts add up to the number of labelled pixels
    big = torch.zeros((4096, 4096), dtype=torch.int32, device='cuda')
    big[100:3000:7, 50:4000:5] = 1
    big[3500:3600, 3500:3700] = 77
    bb, cnt = cutout.segment_bounding_boxes(big)
    assert int(cnt.sum()) == int((big != 0).sum())
    assert bb[77].tolist() == [3500, 3500, 3699, 3599] and int(cnt[77]) == 100 * 200
    assert bb[1].tolist() == [50, 100, 3995, 2998]
