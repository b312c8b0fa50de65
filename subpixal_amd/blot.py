"""Half-pixel dithered blots on the MI355X.

The reference builds the four inputs of ``cc.find_displacement`` by calling
``blot.blot_cutout(dzct, imct)`` four times per source with the image cutout's grid
displaced by half a pixel (/root/reference/subpixal/align.py:664-676); each call runs
drizzlepac's C ``tblot`` with ``interp='poly5'`` through a Python WCS callback
(blot.py:79-155).  drizzlepac and the WCS stack are not part of the reference tree, so
this module does NOT reproduce ``blot_cutout`` itself.  What it provides is the same
resampling for all sources and all four dithers in one kernel launch, for coordinate maps given
per cutout either as an affine (``spx_blot_affine4_f32``: a local linearisation of target-pixel ->
source-pixel) or, where instrument distortion makes the map non-affine at the 1e-3 px level, as a
bivariate polynomial of degree <= 5 (``spx_blot_poly4_f32``); ``map_from`` picks the cheapest form
that reproduces a callable map to a tolerance and refuses when none does.  The quintic interpolant is
restated from its published form (IRAF bipoly5 / Everett's formula), see
``spx_aux_kernels.h``; parity with drizzlepac is unpinned.
"""
import numpy as np
import torch

from . import _ffi, device

__all__ = ['blot_affine4_batch', 'blot_poly4_batch', 'blot4_packed', 'affine_from_map', 'poly_from_map', 'map_from',
           'shift_affine', 'POLY_TERMS']

POLY_TERMS = 21          # monomials u^i v^j with i + j <= 5


def shift_affine(count, x0=0.0, y0=0.0, scale=1.0):
    """``[count, 6]`` affines of a pure offset/scale map: target pixel (x, y) lies at source
    pixel ``(scale*x + x0, scale*y + y0)``; ``x0, y0`` scalars or ``[count]`` arrays."""
    a = np.zeros((int(count), 6), dtype=np.float64)
    a[:, 0] = scale
    a[:, 4] = scale
    a[:, 2] = x0
    a[:, 5] = y0
    return a


def affine_from_map(mapping, shape):
    """Least-squares affine of a coordinate map over one target cutout.

    mapping : callable ``(x, y) -> (xs, ys)`` on 0-based pixel-centre arrays of the
        un-dithered target grid (e.g. ``lambda x, y: drz.world2pix(*img.pix2world(x, y))``,
        the composition BlotWCSMap evaluates at blot.py:71-76, shifted to 0-based).
    shape : ``(ny, nx)`` of the target cutout.
    Returns ``(affine[6], max_residual_px)``: the residual over a 5x5 grid of probe points
    says how well the map is affine over the cutout (distortion shows up here).
    """
    ny, nx = int(shape[0]), int(shape[1])
    gx, gy = np.meshgrid(np.linspace(0.0, nx - 1.0, 5), np.linspace(0.0, ny - 1.0, 5))
    gx, gy = gx.ravel(), gy.ravel()
    xs, ys = mapping(gx, gy)
    design = np.stack([gx, gy, np.ones_like(gx)], axis=1)
    cx, *_ = np.linalg.lstsq(design, np.asarray(xs, dtype=np.float64), rcond=None)
    cy, *_ = np.linalg.lstsq(design, np.asarray(ys, dtype=np.float64), rcond=None)
    res = np.hypot(design @ cx - xs, design @ cy - ys).max()
    return np.concatenate([cx, cy]), float(res)


def _poly_design(u, v, degree):
    """Columns u^i v^(d-i) in the kernel's order: k = d(d+1)/2 + (d - i), d = 0..degree."""
    cols = []
    for d in range(degree + 1):
        for i in range(d, -1, -1):
            cols.append(u ** i * v ** (d - i))
    return np.stack(cols, axis=1)


def poly_from_map(mapping, shape, degree=3):
    """Least-squares bivariate polynomial of a coordinate map over one target cutout, for maps
    that are NOT affine over it (HST FLT distortion is not, at the 1e-3 px level, over 64-128 px):
    what the reference's ``BlotWCSMap`` evaluates per pixel (blot.py:21-76).

    mapping, shape : as :func:`affine_from_map`.
    degree : total degree 1..5.
    Returns ``(coef[2, 21], max_residual_px)``; the polynomial is in ``u = x - (nx-1)/2``,
    ``v = y - (ny-1)/2`` (``spx_blot_poly4_f32``), the residual is taken over a 13x13 probe grid
    (the fit uses a 9x9 one) INCLUDING the half-pixel dithered positions' range.
    """
    degree = int(degree)
    if not 1 <= degree <= 5:
        raise ValueError("degree must be 1..5")
    ny, nx = int(shape[0]), int(shape[1])
    xc, yc = 0.5 * (nx - 1), 0.5 * (ny - 1)

    def grid(k):
        gx, gy = np.meshgrid(np.linspace(0.0, nx - 0.5, k), np.linspace(0.0, ny - 0.5, k))
        return gx.ravel(), gy.ravel()
    gx, gy = grid(9)
    xs, ys = mapping(gx, gy)
    design = _poly_design(gx - xc, gy - yc, degree)
    cx, *_ = np.linalg.lstsq(design, np.asarray(xs, dtype=np.float64), rcond=None)
    cy, *_ = np.linalg.lstsq(design, np.asarray(ys, dtype=np.float64), rcond=None)
    px, py = grid(13)
    pxs, pys = mapping(px, py)
    pd = _poly_design(px - xc, py - yc, degree)
    res = np.hypot(pd @ cx - pxs, pd @ cy - pys).max()
    coef = np.zeros((2, POLY_TERMS))
    coef[0, :len(cx)] = cx
    coef[1, :len(cy)] = cy
    return coef, float(res)


def map_from(mapping, shape, tol=1e-3):
    """The cheapest map form that reproduces ``mapping`` over the cutout to ``tol`` pixels:
    ``('affine', a[6], residual)`` or ``('poly', (coef[2, 21], degree), residual)`` with the lowest
    sufficient degree.  Raises ValueError when even degree 5 leaves a larger residual (the blot
    would be displaced by more than ``tol``: refuse rather than resample wrongly)."""
    a, res = affine_from_map(mapping, shape)
    if res <= tol:
        return 'affine', a, res
    for degree in (2, 3, 4, 5):
        coef, res = poly_from_map(mapping, shape, degree)
        if res <= tol:
            return 'poly', (coef, degree), res
    raise ValueError("coordinate map is not a degree-5 polynomial over the cutout to %g px "
                     "(residual %.3g px)" % (tol, res))


def blot_poly4_batch(src, coef, shape, degree, gain=None):
    """The four dithered blots of every source through a polynomial coordinate map
    (``spx_blot_poly4_f32``): ``coef [N, 2, 21]`` float64 as :func:`poly_from_map` returns them,
    everything else as :func:`blot_affine4_batch`."""
    like_torch = isinstance(src, torch.Tensor)
    s = device.to_device(src, torch.float32)
    if s.dim() != 3:
        raise ValueError("src must have shape [N, sny, snx].")
    c = device.to_device(np.asarray(coef, dtype=np.float64) if not isinstance(coef, torch.Tensor)
                         else coef, torch.float64)
    if c.dim() != 3 or tuple(c.shape[1:]) != (2, POLY_TERMS) or c.shape[0] != s.shape[0]:
        raise ValueError("coef must have shape [N, 2, 21].")
    g = None
    if gain is not None:
        g = device.to_device(np.asarray(gain, dtype=np.float32) if not isinstance(gain, torch.Tensor)
                             else gain, torch.float32)
        if g.dim() != 1 or g.shape[0] != s.shape[0]:
            raise ValueError("gain must have shape [N].")
    ny, nx = int(shape[0]), int(shape[1])
    im4 = torch.empty((s.shape[0], 4, ny, nx), dtype=torch.float32, device=s.device)
    lib = _ffi.load()
    with torch.cuda.device(s.device):
        _ffi.check(lib.spx_blot_poly4_f32(device.ptr(s), s.shape[0], s.shape[1], s.shape[2],
                                          device.ptr(c), int(degree), device.ptr(g), ny, nx,
                                          device.ptr(im4), device.stream_ptr()))
    return im4 if like_torch else im4.cpu().numpy()


def blot_affine4_batch(src, affine, shape, gain=None):
    """The four dithered blots of every source.

    src : ``[N, sny, snx]`` float32 drizzled cutouts (masked pixels already zero,
        align.py:661), torch CUDA tensor or numpy array; sny, snx >= 6.
    affine : ``[N, 6]`` float64, target pixel ``(x, y)`` -> source pixel
        ``(a0 x + a1 y + a2, a3 x + a4 y + a5)`` for the un-dithered target grid.
    shape : ``(ny, nx)`` of the target (image) cutouts.
    gain : optional ``[N]`` factor on the samples (blot.py:134-150's exposure-time and
        pixel-area scaling, which depends on header values the caller holds).

    Returns ``im4 [N, 4, ny, nx]`` float32 (image00, image10, image01, image11 as
    ``find_displacement_batch`` takes them): dither (ox, oy) in {0, 1/2}^2 (the reference's
    ``imct.dx -= ox; imct.dy -= oy``) samples the target position ``(x + ox, y + oy)``
    (cutout.py:1138); points mapping outside the source are 0.
    Torch tensor if ``src`` was one, numpy otherwise.
    """
    like_torch = isinstance(src, torch.Tensor)
    s = device.to_device(src, torch.float32)
    if s.dim() != 3:
        raise ValueError("src must have shape [N, sny, snx].")
    a = device.to_device(np.asarray(affine, dtype=np.float64) if not isinstance(affine, torch.Tensor)
                         else affine, torch.float64)
    if a.dim() != 2 or a.shape[1] != 6 or a.shape[0] != s.shape[0]:
        raise ValueError("affine must have shape [N, 6].")
    g = None
    if gain is not None:
        g = device.to_device(np.asarray(gain, dtype=np.float32) if not isinstance(gain, torch.Tensor)
                             else gain, torch.float32)
        if g.dim() != 1 or g.shape[0] != s.shape[0]:
            raise ValueError("gain must have shape [N].")
    ny, nx = int(shape[0]), int(shape[1])
    im4 = torch.empty((s.shape[0], 4, ny, nx), dtype=torch.float32, device=s.device)
    lib = _ffi.load()
    with torch.cuda.device(s.device):
        _ffi.check(lib.spx_blot_affine4_f32(device.ptr(s), s.shape[0], s.shape[1], s.shape[2],
                                            device.ptr(a), device.ptr(g), ny, nx, device.ptr(im4),
                                            device.stream_ptr()))
    return im4 if like_torch else im4.cpu().numpy()


def blot4_packed(src, src_offsets, src_shapes, maps, dst_offsets, dst_shapes, dst_total, degree=0, gain=None):
    """The four dithered blots of every source of a packed catalog (``spx_blot4_var_f32``): ``src`` holds the
    drizzled cutouts back to back (``cutout.pack_cutouts_var`` layout: ``src_offsets`` int64 [N],
    ``src_shapes`` int32 [N, 2] = (h, w)), the blots are made for image cutouts laid out by ``dst_offsets`` /
    ``dst_shapes`` (``dst_total`` pixels in all).  ``maps``: ``[N, 6]`` affines (``degree`` 0) or ``[N, 2, 21]``
    polynomial coefficients (``degree`` 1..5).  Returns the flat float32 device buffer
    ``[item][4][pixels]`` (item k at ``4 * dst_offsets[k]``)."""
    n = int(src_offsets.shape[0])
    degree = int(degree)
    m = np.asarray(maps, dtype=np.float64) if not isinstance(maps, torch.Tensor) else maps
    want = (n, 6) if degree == 0 else (n, 2, POLY_TERMS)
    if tuple(m.shape) != want:
        raise ValueError("maps must have shape %s." % (want,))
    m = device.to_device(m, torch.float64)
    g = None
    if gain is not None:
        g = device.to_device(np.asarray(gain, dtype=np.float32) if not isinstance(gain, torch.Tensor)
                             else gain, torch.float32)
        if g.dim() != 1 or g.shape[0] != n:
            raise ValueError("gain must have shape [N].")
    im4 = torch.empty((max(4 * int(dst_total), 1),), dtype=torch.float32, device=src.device)
    lib = _ffi.load()
    with torch.cuda.device(src.device):
        _ffi.check(lib.spx_blot4_var_f32(device.ptr(src), device.ptr(src_offsets), device.ptr(src_shapes), n,
                                         device.ptr(m), degree, device.ptr(g), device.ptr(dst_offsets),
                                         device.ptr(dst_shapes), device.ptr(im4), device.stream_ptr()))
    return im4
