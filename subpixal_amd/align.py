"""Batched counterpart of the reference's per-source fit loop.

``find_linear_fit`` keeps the name, arguments and return triple of
``subpixal.align.find_linear_fit`` (/root/reference/subpixal/align.py:561-745) but runs
the loop body align.py:656-699 -- zero the masked pixels, take the four half-pixel
blots, ``cc.find_displacement`` -- for ALL sources in one kernel launch per cutout
shape (the iterations are independent; nothing computed for source k feeds k+1 until
the linear fit at align.py:720).

Not reproduced here (out of scope): the blotting itself
(``blot.blot_cutout`` -> drizzlepac's C ``tblot``) and astropy WCS objects.  The caller
supplies the dithered blots, either ready-made or through a ``blot`` callable with the
reference's ``blot_cutout(dzct, imct)`` signature, and -- optionally -- WCS objects with the
astropy methods the reference calls (duck-typed).

``iter_linear_fit`` stands in for ``tweakwcs.linearfit.iter_linear_fit`` (align.py:720-724):
tweakwcs is not available here, so its numerical parity is UNPINNED; the implementation follows
the documented behaviour (weighted 'shift' / 'rscale' / 'general' fits about a centre with
iterative sigma clipping).  It is a few-thousand-point 2x3 least squares and stays on the host.
"""
import numpy as np

from . import _ffi, cc

__all__ = ['find_linear_fit', 'iter_linear_fit', 'measure_shifts', 'measure_shifts_affine',
           'usable_status', 'ST_SKIPPED']

# per-source status of the measurement: the library's SPX_ST_* codes (include/subpixal_hip.h) or
ST_SKIPPED = -1          # cutout shape outside what the kernels take (3.._ffi.MAX_SIDE = 682 px per side): not measured


def usable_status(status):
    """True where a displacement is one the reference itself would have produced from finite data:
    accepted vertex (0) or one of ``find_peak``'s own fallbacks (edge 1, no maximum 2, vertex outside 3:
    centroid.py:171-172, 218-236).  Non-finite cutouts (6), an unbracketed refinement window (4) and
    skipped shapes (-1) carry no measurement; ``find_linear_fit`` gives them zero weight."""
    status = np.asarray(status)
    return (status >= _ffi.ST_OK) & (status <= _ffi.ST_OUTSIDE)


def _shape_ok(shape):
    return len(shape) == 2 and all(3 <= n <= _ffi.MAX_SIDE for n in shape)


# ----------------------------------------------------------------------------
# robust linear fit of matched point lists (host, float64)
# ----------------------------------------------------------------------------
def _weighted_fit(xy, uv, w, fitgeom):
    """Fit uv ~ F @ xy + t (xy already centred) with weights ``w`` (zero = point not used; no copies of the
    point lists are made, so a clipping round costs a few passes over the arrays).  Returns (F 2x2, t 2)."""
    sw = np.sum(w)
    if fitgeom == 'shift':
        t = w @ (uv - xy) / sw
        return np.eye(2), t
    mx = w @ xy / sw
    mu = w @ uv / sw
    x = xy - mx
    u = uv - mu
    if fitgeom == 'rscale':
        # similarity transform: u = s R x  (least squares, closed form)
        sxx = np.sum(w * (x[:, 0] ** 2 + x[:, 1] ** 2))
        a = np.sum(w * (x[:, 0] * u[:, 0] + x[:, 1] * u[:, 1])) / sxx
        b = np.sum(w * (x[:, 0] * u[:, 1] - x[:, 1] * u[:, 0])) / sxx
        f = np.array([[a, -b], [b, a]])
    elif fitgeom == 'general':
        # affine: normal equations on the centred coordinates (a 2x2 system whose condition number is the
        # aspect ratio of the point cloud): F = (sum w u x^T) (sum w x x^T)^-1
        xw = x * w[:, None]
        f = np.linalg.solve(xw.T @ x, xw.T @ u).T
    else:
        raise ValueError("Unsupported 'fitgeom'. Valid values are: 'shift', 'rscale', 'general'.")
    t = mu - f @ mx
    return f, t


def iter_linear_fit(xy, uv, wxy=None, wuv=None, fitgeom='general', center=None, nclip=3,
                    sigma=3.0):
    """Iteratively sigma-clipped weighted linear fit ``uv ~ F (xy - c) + c + t``.

    Returns a dict with ``offset`` (t), ``fit_matrix`` (F), ``rot`` / ``scale`` / ``skew`` derived from
    F, ``rms`` (per axis), ``resids``, ``fitmask`` (points kept), ``eff_nclip``, ``center``.
    """
    xy = np.asarray(xy, dtype=np.float64)
    uv = np.asarray(uv, dtype=np.float64)
    if xy.shape != uv.shape or xy.ndim != 2 or xy.shape[1] != 2:
        raise ValueError("Input coordinate lists must both have shape (N, 2).")
    n = xy.shape[0]
    minpts = {'shift': 1, 'rscale': 2, 'general': 3}.get(fitgeom)
    if minpts is None:
        raise ValueError("Unsupported 'fitgeom'. Valid values are: 'shift', 'rscale', 'general'.")
    if n < minpts:
        raise ValueError("Not enough points for the requested fit geometry.")
    w = np.ones(n)
    for extra in (wxy, wuv):
        if extra is not None:
            w = w * np.asarray(extra, dtype=np.float64)
    c = np.zeros(2) if center is None else np.asarray(center, dtype=np.float64)
    mask = w > 0.0 if np.any(w == 0.0) else np.ones(n, dtype=bool)     # zero weight = not a measurement
    if mask.sum() < minpts:
        raise ValueError("Not enough points with non-zero weight for the requested fit geometry.")
    eff = 0
    xc, uc = xy - c, uv - c
    for it in range(max(0, int(nclip)) + 1):
        wm = w * mask
        f, t = _weighted_fit(xc, uc, wm, fitgeom)
        resid = uc - (xc @ f.T + t)
        if it == nclip or sigma is None:
            break
        r2 = resid[:, 0] ** 2 + resid[:, 1] ** 2
        rms = np.sqrt((wm @ r2) / np.sum(wm))
        keep = mask & (r2 <= (sigma * rms) ** 2)
        nkeep = np.count_nonzero(keep)
        if nkeep == np.count_nonzero(mask) or nkeep < minpts:
            break
        mask = keep
        eff += 1
    wm = w * mask
    rms = np.sqrt((wm @ resid ** 2) / np.sum(wm))
    rotx = np.degrees(np.arctan2(f[1, 0], f[0, 0]))
    roty = np.degrees(np.arctan2(-f[0, 1], f[1, 1]))
    sx = float(np.hypot(f[0, 0], f[1, 0]))
    sy = float(np.hypot(f[0, 1], f[1, 1]))
    return {
        'offset': t, 'shift': t, 'fit_matrix': f, 'matrix': f,
        'rot': 0.5 * (rotx + roty), 'rotxy': (rotx, roty, 0.5 * (rotx + roty), roty - rotx),
        'scale': (np.sqrt(abs(np.linalg.det(f))), sx, sy), 'skew': roty - rotx,
        'rms': rms, 'resids': resid, 'fitmask': mask, 'eff_nclip': eff, 'center': c,
        'fitgeom': fitgeom, 'nmatches': int(mask.sum()),
    }


# ----------------------------------------------------------------------------
# the batched loop body of align.py:656-699
# ----------------------------------------------------------------------------
def measure_shifts(ref_tiles, im4_tiles, cc_type='NCC', full_output=False, return_status=False):
    """Displacements for lists of same-or-mixed-shape cutouts: ``ref_tiles[k]`` is a 2-D
    array, ``im4_tiles[k]`` its four dithered blots (00, 10, 01, 11).  One launch per kernel
    family (``cc.find_displacement_var``), not per shape.  Returns ``dxdy [N, 2]`` (then the list of interlaced images with
    ``full_output``, then ``status [N]`` with ``return_status``).  A cutout whose shape the kernels
    do not take (outside 3..682 px per side; 129..682 px run on the general path) is not measured: shift 0,
    status ST_SKIPPED --
    one oversized source must not abort the whole fit."""
    for k, r in enumerate(ref_tiles):
        shapes = {np.shape(r)} | {np.shape(b) for b in im4_tiles[k]}
        if len(shapes) != 1:
            raise ValueError("All cutouts must have same shape.")       # cc.py:103-105
    dxdy, iccs, status = cc.find_displacement_var(ref_tiles, im4_tiles, cc_type=cc_type, full_output=True,
                                                  return_status=True)
    dxdy = np.where(status[:, None] == ST_SKIPPED, 0.0, dxdy)          # not measured: shift 0, zero weight
    out = [dxdy]
    if full_output:
        out.append(iccs)
    if return_status:
        out.append(status)
    return out[0] if len(out) == 1 else tuple(out)


def _image_xy(ct, x, y, wcslin):
    """Position (x, y) of cutout ``ct`` in the tangent-plane image coordinates of ``wcslin``
    (align.py:692-699); without WCS objects, plain pixel coordinates of the parent image."""
    if wcslin is not None and getattr(ct, 'wcs', None) is not None:
        ra, dec = ct.pix2world(x, y)
        return np.array(wcslin.wcs_world2pix(ra, dec, 1), dtype=np.float64).reshape(2)
    return np.array([x + ct.blc[0] - ct.dx + 1.0, y + ct.blc[1] - ct.dy + 1.0])


def measure_shifts_affine(img_tiles, drz_tiles, affine, gain=None, cc_type='NCC', degree=None):
    """The loop body of align.py:656-689 with the four blots made on the GPU
    (``blot.blot_affine4_batch``): ``img_tiles[k]`` and ``drz_tiles[k]`` are 2-D arrays,
    ``affine[k]`` maps image-cutout pixels to drizzled-cutout pixels (``[N, 6]``; with ``degree``
    given it is the ``[N, 2, 21]`` polynomial coefficients of ``blot.poly_from_map`` instead and
    ``blot.blot_poly4_batch`` does the resampling).  The blots never leave the device.  Returns ``(dxdy [N, 2], interlaced images, non-shifted blots, status [N])``;
    shapes the kernels do not take are skipped as in :func:`measure_shifts`."""
    import torch
    from . import blot as _blot
    n = len(img_tiles)
    affine = np.asarray(affine, dtype=np.float64)
    affine = affine.reshape(n, 6) if degree is None else affine.reshape(n, 2, _blot.POLY_TERMS)
    gain = None if gain is None else np.asarray(gain, dtype=np.float32).reshape(n)
    dxdy = np.zeros((n, 2), dtype=np.float64)
    status = np.full(n, ST_SKIPPED, dtype=np.int32)
    iccs, blt00 = [None] * n, [None] * n
    groups = {}
    for k in range(n):
        if _shape_ok(np.shape(img_tiles[k])) and min(np.shape(drz_tiles[k])) >= 6:
            groups.setdefault((np.shape(img_tiles[k]), np.shape(drz_tiles[k])), []).append(k)
    for (ishape, _), idx in groups.items():
        ref = torch.as_tensor(np.stack([np.asarray(img_tiles[k], dtype=np.float32) for k in idx])).cuda()
        src = torch.as_tensor(np.stack([np.asarray(drz_tiles[k], dtype=np.float32) for k in idx])).cuda()
        g = None if gain is None else gain[idx]
        im4 = _blot.blot_affine4_batch(src, affine[idx], ishape, g) if degree is None else \
            _blot.blot_poly4_batch(src, affine[idx], ishape, degree, g)
        d, icc, st = cc.find_displacement_batch(ref, im4, cc_type=cc_type, full_output=True,
                                                return_status=True)
        d, icc, b0 = d.cpu().numpy(), icc.cpu().numpy(), im4[:, 0].cpu().numpy()
        dxdy[idx] = d
        status[idx] = st.cpu().numpy()
        for j, k in enumerate(idx):
            iccs[k] = icc[j]
            blt00[k] = b0[j]
    return dxdy, iccs, blt00, status


def _fit_from_shifts(img_dxy, status, xyim, xyref, weights, wcslin, fitgeom, nclip, sigma):
    """The tail of align.py:701-745 shared by the list and the catalog path: zero weight for sources
    without a measurement, the robust fit, ``irmse``."""
    npts = len(img_dxy)
    good = usable_status(status)
    ref_dxy = xyim - xyref
    user_weights = weights is not None
    if not np.all(good):
        weights = (np.ones(npts) if weights is None else weights) * good
    center = None
    if wcslin is not None and hasattr(wcslin, 'wcs'):
        center = np.array(wcslin.wcs.crpix)
    fit = iter_linear_fit(xyim, xyref, wxy=None, wuv=weights, fitgeom=fitgeom, center=center,
                          nclip=nclip, sigma=sigma)
    fit['subpixal_img_dxy'] = img_dxy
    fit['subpixal_ref_dxy'] = ref_dxy
    fit['subpixal_status'] = status
    m = fit['fitmask']
    if not user_weights:                                                   # align.py:730-743
        fit['irmse'] = float(np.sqrt(2 * np.mean(img_dxy[m] ** 2)))
    else:
        wt = np.sum(weights)
        if len(weights) == 0 or wt == 0.0:
            fit['irmse'] = float('nan')
        else:
            w = weights / wt
            fit['irmse'] = float(np.sqrt(np.sum(np.dot(w[m], img_dxy[m] ** 2))))
    return fit


def _find_linear_fit_catalog(img_cat, drz_cat, wcslin, fitgeom, nclip, sigma, use_weights, cc_type, blot,
                             affine, gain, poly):
    """``find_linear_fit`` for two :class:`~subpixal_amd.cutout.CutoutCatalog` (the image's and the drizzled
    image's cutouts of the same sources): the loop body of align.py:656-699 runs on the device for the whole
    catalog -- gather of the variable-shape cutouts from the resident frames, masked pixels of the drizzled
    cutouts zeroed (align.py:661), the four blots per source, ``cc.find_displacement`` -- with no host loop
    over sources; only the (dx, dy) and status arrays come back for the fit.  Per source the arithmetic is that
    of the list path, so the shifts are bit-identical to calling ``cc.find_displacement`` source by source."""
    import torch
    from . import blot as _blot
    from .cutout import CutoutCatalog, PackedImages
    if not (isinstance(img_cat, CutoutCatalog) and isinstance(drz_cat, CutoutCatalog)):
        raise ValueError("img_cutouts and drz_cutouts must both be CutoutCatalog objects (or neither).")
    if len(img_cat) != len(drz_cat):                                       # align.py:631-633
        raise ValueError("The number of image cutouts must match the number "
                         "of drizzled cutouts.")
    if blot is not None or (affine is None) == (poly is None):
        raise ValueError("The catalog path takes the target->source maps as 'affine' or 'poly' (blot.map_from), "
                         "not a 'blot' callable.")
    npts = len(img_cat)
    degree = 0
    maps = affine
    if poly is not None:
        maps, degree = poly
    if wcslin is None and drz_cat.wcs is not None:
        wcslin = drz_cat.wcs                                               # align.py:636-639
    if wcslin is not None and hasattr(wcslin, 'deepcopy'):
        wcslin = wcslin.deepcopy()

    img_p, img_off, img_shp = img_cat.packed(zero_masked=False)
    drz_p, drz_off, drz_shp = drz_cat.packed(zero_masked=True)             # align.py:661
    shapes = img_cat.shapes
    total = int((shapes[:, 0].astype(np.int64) * shapes[:, 1]).sum())
    im4 = _blot.blot4_packed(drz_p, drz_off, drz_shp, maps, img_off, img_shp, total, degree, gain)
    d, st, icc = cc.find_displacement_packed(img_p, im4, img_off, img_shp, shapes, cc_type=cc_type)
    offs_host = np.zeros(npts, dtype=np.int64)
    if npts > 1:
        np.cumsum((shapes[:-1, 0].astype(np.int64) * shapes[:-1, 1]), out=offs_host[1:])
    # cutouts above 128 px (general path): one launch per shape, from the packed buffers (rare)
    side, low = shapes.max(axis=1), shapes.min(axis=1)
    for shp in {tuple(x) for x in shapes[(side > 128) & (side <= _ffi.MAX_SIDE) & (low >= 3)]}:
        idx = np.nonzero((shapes == np.array(shp)).all(axis=1))[0]
        npx = shp[0] * shp[1]
        ref = torch.stack([img_p[o:o + npx].view(*shp) for o in offs_host[idx]])
        b4 = torch.stack([im4[4 * o:4 * o + 4 * npx].view(4, *shp) for o in offs_host[idx]])
        db, ib, sb = cc.find_displacement_batch(ref, b4, cc_type=cc_type, full_output=True, return_status=True)
        it = torch.from_numpy(idx).to(d.device)
        d[it], st[it] = db, sb
        for j, o in enumerate(offs_host[idx]):
            icc[4 * o:4 * o + 4 * npx] = ib[j].reshape(-1)
    img_dxy = d.cpu().numpy()
    status = st.cpu().numpy()
    img_dxy = np.where(status[:, None] == ST_SKIPPED, 0.0, img_dxy)        # not measured: shift 0, zero weight

    def frame_xy(x, y):
        # align.py:692-699 for every source at once; without WCS objects: 1-based pixel coordinates of the
        # parent image (the list path's `_image_xy`)
        if wcslin is not None and img_cat.wcs is not None:
            ra, dec = img_cat.wcs.all_pix2world(x, y, 0)
            return np.stack(wcslin.wcs_world2pix(ra, dec, 1), axis=1).astype(np.float64)
        return np.stack([x + 1.0, y + 1.0], axis=1)

    xyim = frame_xy(img_cat.src_pos[:, 0], img_cat.src_pos[:, 1])
    xyref = frame_xy(img_cat.src_pos[:, 0] + img_dxy[:, 0], img_cat.src_pos[:, 1] + img_dxy[:, 1])
    weights = drz_cat.src_weight if use_weights else None                  # align.py:703-716
    fit = _fit_from_shifts(img_dxy, status, xyim, xyref, weights, wcslin, fitgeom, nclip, sigma)
    interlaced_cc = PackedImages(icc, offs_host, 2 * shapes, scale=4)
    nonshifted_blts = PackedImages(im4, offs_host, shapes, scale=4, part=0)
    return fit, interlaced_cc, nonshifted_blts


def find_linear_fit(img_cutouts, drz_cutouts, wcslin=None, fitgeom='general',
                    nclip=3, sigma=3.0, use_weights=True, cc_type='NCC', blot=None,
                    affine=None, gain=None, poly=None):
    """Linear fit to the displacements (found by cross-correlation) between ``img_cutouts`` and
    the blots of ``drz_cutouts`` onto them.  Same arguments and return value
    ``(fit, interlaced_cc, nonshifted_blts)`` as the reference (align.py:561-745).

    blot : callable ``blot(dzct, imct) -> cutout`` with the semantics of the reference's
        ``blot_cutout``; called four times per source with the image cutout's grid displaced by
        (0,0), (-1/2,0), (-1/2,-1/2), (0,-1/2) exactly as align.py:664-679.  When None, each element
        of ``drz_cutouts`` must already be the 4-sequence ``(blt00, blt10, blt01, blt11)``.
    affine : ``[N, 6]`` image-cutout pixel -> drizzled-cutout pixel maps (``blot.affine_from_map``);
        when given, ``drz_cutouts`` are the drizzled cutouts themselves and the four blots are
        resampled on the GPU (``blot.blot_affine4_batch``, quintic interpolation), with the optional
        per-source ``gain`` of blot.py:134-150.  ``blot`` must be None then.
    poly : ``(coef [N, 2, 21], degree)`` polynomial maps (``blot.poly_from_map`` / ``blot.map_from``) for
        cutouts over which instrument distortion makes the map non-affine; used like ``affine``.
    """
    from .cutout import CutoutCatalog
    if isinstance(img_cutouts, CutoutCatalog) or isinstance(drz_cutouts, CutoutCatalog):
        return _find_linear_fit_catalog(img_cutouts, drz_cutouts, wcslin, fitgeom, nclip, sigma, use_weights,
                                        cc_type, blot, affine, gain, poly)
    if not hasattr(img_cutouts, '__iter__'):
        img_cutouts = [img_cutouts]
    if not hasattr(drz_cutouts, '__iter__') or (blot is not None and hasattr(drz_cutouts, 'data')):
        drz_cutouts = [drz_cutouts]
    img_cutouts = list(img_cutouts)
    drz_cutouts = list(drz_cutouts)
    if len(img_cutouts) != len(drz_cutouts):                               # align.py:631-633
        raise ValueError("The number of image cutouts must match the number "
                         "of drizzled cutouts.")
    npts = len(img_cutouts)
    if wcslin is None and blot is not None and getattr(drz_cutouts[0], 'wcs', None) is not None:
        wcslin = drz_cutouts[0].wcs                                        # align.py:636-639
    if wcslin is not None and hasattr(wcslin, 'deepcopy'):
        wcslin = wcslin.deepcopy()

    if sum(x is not None for x in (affine, blot, poly)) > 1:
        raise ValueError("Give either a 'blot' callable or 'affine' maps or 'poly' maps, not several.")
    degree = None
    if poly is not None:
        affine, degree = poly

    def data_of(c):
        return c.data if hasattr(c, 'data') and not isinstance(c, np.ndarray) else np.asarray(c)

    blts = []
    for imct, dz in zip(img_cutouts, drz_cutouts):
        if affine is not None:
            if hasattr(dz, 'mask') and not isinstance(dz, np.ndarray):
                dz.data[dz.mask] = 0                                       # align.py:661
            continue
        if blot is None:
            if len(dz) != 4:
                raise ValueError("Without a 'blot' callable each element of drz_cutouts must be "
                                 "the four dithered blots (blt00, blt10, blt01, blt11).")
            blts.append(tuple(dz))
            continue
        dx0, dy0 = imct.dx, imct.dy                                        # align.py:658-679
        dz.data[dz.mask] = 0
        b00 = blot(dz, imct)
        imct.dx -= 0.5
        b10 = blot(dz, imct)
        imct.dy -= 0.5
        b11 = blot(dz, imct)
        imct.dx = dx0
        b01 = blot(dz, imct)
        imct.dy = dy0
        blts.append((b00, b10, b01, b11))

    if affine is not None:
        img_dxy, interlaced_cc, nonshifted_blts, status = measure_shifts_affine(
            [data_of(c) for c in img_cutouts], [data_of(c) for c in drz_cutouts], affine, gain,
            cc_type=cc_type, degree=degree)
    else:
        img_dxy, interlaced_cc, status = measure_shifts(
            [data_of(c) for c in img_cutouts],
            [[data_of(b) for b in four] for four in blts], cc_type=cc_type, full_output=True,
            return_status=True)
        nonshifted_blts = [four[0] for four in blts]
    xyim = np.empty((npts, 2))
    xyref = np.empty((npts, 2))
    for k, imct in enumerate(img_cutouts):
        if hasattr(imct, 'cutout_src_pos'):
            x1, y1 = imct.cutout_src_pos
            xyim[k] = _image_xy(imct, x1, y1, wcslin)
            xyref[k] = _image_xy(imct, x1 + img_dxy[k, 0], y1 + img_dxy[k, 1], wcslin)
        else:                       # bare arrays: positions are the cutout centres
            ny, nx = np.shape(data_of(imct))
            xyim[k] = ((nx - 1) / 2.0, (ny - 1) / 2.0)
            xyref[k] = xyim[k] + img_dxy[k]

    weights = None
    if use_weights:                                                        # align.py:703-716
        carriers = drz_cutouts if (blot is not None or affine is not None) else [four[0] for four in blts]
        weights = [getattr(c, 'src_weight', None) for c in carriers]
        if all(w is None for w in weights):
            weights = None
        elif any(w is None for w in weights):
            raise ValueError("Not all cutouts have weights set. All cutouts "
                             "must either have non-negative weights or be "
                             "None.")
        elif any(w < 0 for w in weights):
            raise ValueError("Weights must be non-negative.")
        else:
            weights = np.asarray(weights, dtype=np.float64)
    # Sources without a measurement (non-finite pixels, e.g. the NaN fill of a cutout overhanging
    # its frame; an oversized cutout) get zero weight instead of poisoning the fit with a
    # meaningless shift.  The reference has no such guard: it would fit whatever came back.
    fit = _fit_from_shifts(img_dxy, status, xyim, xyref, weights, wcslin, fitgeom, nclip, sigma)
    return fit, interlaced_cc, nonshifted_blts
