#!/usr/bin/env python
"""BASELINE.json config 5 through the API surface the north star keeps: `subpixal_amd.align.find_linear_fit`
on real cutout carriers with VARIABLE shapes (bounding box + padding per source, cutout.py:159-175) and the
reference's 5-image `cc.find_displacement` (align.py:656-699), for a synthetic 4096x4096 frame pair with a
5000-source catalog -- device-resident: the frames live on the GPU, `cutout.CutoutCatalog` describes the
cutouts by a box table, and the whole per-source loop is four kernel launches.

    python tools/align_catalog.py [--size 4096] [--sources 5000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import align_synthetic                                      # noqa: E402  (scene + renderer)


def build(size=4096, nsrc=5000, seed=5, margin=6, pad=3):
    """Frames, segmentation, catalogs and maps.

    Sources sit on a jittered grid (no two closer than ~35 px), 98 % compact (sigma 4 px, segment = ellipse
    with semi-axes of 11..17 px) and 2 % extended (sigma 9..13 px, semi-axes 30..45 px, which swallow
    neighbours).  The drizzled
    frame holds them at xy, the image frame at xy2 = T(xy).  As in the reference, the primary cutouts are the
    segments' bounding boxes + `pad` (cutout.py:138-175; `cutout.primary_cutout_boxes` on the GPU) -- one shape
    per source --, the image cutout of a source is the same box moved by the integer part of the displacement,
    the drizzled cutout is the box grown by `margin`, so the image-cutout -> drizzled-cutout pixel map is a pure
    offset; pixels of a drizzled cutout outside its own segment are masked and zeroed (cutout.py:190,
    align.py:661)."""
    import torch
    from subpixal_amd import blot, cutout
    rng = np.random.default_rng(seed)
    g = int(np.ceil(np.sqrt(nsrc)))
    cell = (size - 160.0) / g
    cells = rng.choice(g * g, nsrc, replace=False)
    xy = np.stack([80.0 + (cells % g + 0.5) * cell, 80.0 + (cells // g + 0.5) * cell], axis=1)
    xy += rng.uniform(-0.17, 0.17, (nsrc, 2)) * cell
    amp = rng.uniform(0.5, 2.0, nsrc)
    big = rng.random(nsrc) < 0.02
    rx = np.where(big, rng.integers(30, 46, nsrc), rng.integers(11, 18, nsrc))
    ry = np.where(big, rng.integers(30, 46, nsrc), rng.integers(11, 18, nsrc))
    radius = np.maximum(rx, ry)
    sigma = np.where(big, np.minimum(rx, ry) / 3.4, 4.0)
    c = np.array([size / 2.0, size / 2.0])
    th = np.radians(0.002)
    f = 1.00003 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    f[0, 1] += 1e-5
    t = np.array([0.731, -1.284])
    xy2 = (xy - c) @ f.T + c + t
    drz_frame = np.zeros((size, size), np.float32)
    img_frame = np.zeros((size, size), np.float32)
    seg = np.zeros((size, size), np.int32)
    for k in np.argsort(-radius, kind='stable'):              # extended sources first: compact ones on top
        for frame, (x, y) in ((drz_frame, xy[k]), (img_frame, xy2[k])):
            half = int(4.5 * sigma[k])
            gg = np.arange(-half, half + 1)
            ix, iy = int(round(x)), int(round(y))
            gx = np.exp(-((gg + ix - x) ** 2) / (2 * sigma[k] ** 2))
            gy = np.exp(-((gg + iy - y) ** 2) / (2 * sigma[k] ** 2))
            frame[iy - half:iy + half + 1, ix - half:ix + half + 1] += (amp[k] * np.outer(gy, gx)).astype(np.float32)
        ax, ay = int(rx[k]), int(ry[k])
        gx, gy = np.arange(-ax, ax + 1), np.arange(-ay, ay + 1)
        ix, iy = int(round(xy[k, 0])), int(round(xy[k, 1]))
        inside = (gy[:, None] / ay) ** 2 + (gx[None, :] / ax) ** 2 <= 1.0
        seg[iy - ay:iy + ay + 1, ix - ax:ix + ax + 1][inside] = k + 1
    seg_d = torch.from_numpy(seg).cuda()
    ids, boxes = cutout.primary_cutout_boxes(seg_d, pad=pad)            # 8f-3: one pass over the label image
    k = ids - 1                                                          # (a segment can vanish under others)
    shift = np.round(xy2[k] - xy[k]).astype(np.int32)
    iboxes = boxes.copy()
    iboxes[:, :2] += shift
    dboxes = boxes + np.array([-margin, -margin, 2 * margin, 2 * margin], np.int32)
    weights = rng.uniform(0.5, 2.0, len(k))
    img_cat = cutout.CutoutCatalog(torch.from_numpy(img_frame).cuda(), iboxes, src_pos=xy2[k], src_id=ids)
    drz_cat = cutout.CutoutCatalog(torch.from_numpy(drz_frame).cuda(), dboxes, src_pos=xy[k], src_weight=weights,
                                   src_id=ids, segmentation_image=seg_d)
    img_cat._frame_host, drz_cat._frame_host = img_frame, drz_frame      # (saves the D2H copy when cutouts are built)
    affine = blot.shift_affine(len(k), x0=(shift[:, 0] + margin).astype(np.float64),
                               y0=(shift[:, 1] + margin).astype(np.float64))
    return dict(img_cat=img_cat, drz_cat=drz_cat, affine=affine, xy=xy[k], xy2=xy2[k], f=f, t=t, c=c, compact=~big[k])


def run(size=4096, nsrc=5000, reps=5, quiet=False, nclip=12, cc_type='NCC'):
    import torch
    from subpixal_amd.align import find_linear_fit, iter_linear_fit
    s = build(size, nsrc)
    times = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit, iccs, blts = find_linear_fit(s['img_cat'], s['drz_cat'], affine=s['affine'], fitgeom='general',
                                          nclip=nclip, sigma=3.0, cc_type=cc_type)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    # the transform the fit should find: image positions (1-based) -> drizzled positions, exactly
    exact = iter_linear_fit(s['xy2'] + 1.0, s['xy'] + 1.0, fitgeom='general', nclip=0)
    d = fit['subpixal_img_dxy']
    err = np.abs(d - (s['xy'] - s['xy2'])).max(axis=1)
    out = dict(fit=fit, exact=exact, err=err, first_s=times[0], warm_s=float(np.median(times[1:])),
               scene=s, iccs=iccs, blts=blts)
    nsrc = len(d)
    if not quiet:
        shp = s['img_cat'].shapes
        print('sources %d, frame %dx%d, %d distinct cutout shapes (%d..%d px per side), %s'
              % (nsrc, size, size, len({tuple(x) for x in shp}), shp.min(), shp.max(), cc_type))
        print('find_linear_fit(CutoutCatalog, CutoutCatalog): first call %.2f ms, warm %.2f ms (median of %d)'
              % (1e3 * times[0], 1e3 * out['warm_s'], reps))
        print('|shift - truth|: median %.2e, 90%% %.2e px (compact sources: 99%% %.2e); kept %d of %d'
              % (np.median(err), np.percentile(err, 90), np.percentile(err[s['compact']], 99), fit['fitmask'].sum(), nsrc))
        print('offset err %s, matrix err %.2e' % (fit['offset'] - exact['offset'],
                                                   np.abs(fit['fit_matrix'] - exact['fit_matrix']).max()))
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--sources', type=int, default=5000)
    a = ap.parse_args()
    run(a.size, a.sources)
