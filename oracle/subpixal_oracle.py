"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU (numpy/scipy) restatement of the reference hot path
``subpixal.cc`` + ``subpixal.centroid`` (+ ``utils.py2round``), written from
the reference's behaviour, not copied from it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the checker / the reported CPU baseline.  The
product package ``subpixal_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so this restatement is pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/gen_goldens.py`` (which imports ``/root/reference/subpixal/cc.py``
and ``centroid.py`` in place) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every one of them.

Third-party arithmetic under the path (same libraries the reference calls,
versions of this image): ``scipy.signal.fftconvolve`` (scipy 1.15.3, pocketfft)
at the reference call sites cc.py:114-117, ``numpy.linalg.lstsq`` /
``numpy.argmax`` (numpy 2.2.6) at centroid.py:114,207.

Two groups of functions:

* reference semantics (5-image interlaced mode): ``py2round``, ``find_peak``,
  ``normalize``, ``xcorr_same``, ``build_icc``, ``find_displacement``;
* the pair / ``upsample=U`` mode BASELINE.json asks for, which the reference
  does not have: ``upsampled_cc``, ``xcorr_refine``.
  It is anchored on the reference at two points: U=1 is exactly
  ``fftconvolve(...,'same')[::-1, ::-1] -> find_peak -> - (n-1)//2`` and U=2
  reproduces ``find_displacement`` on band-limited half-pixel dithers.
"""
import numpy as np
from scipy import signal

__all__ = [
    'py2round', 'find_peak', 'find_peak_5x5_all', 'normalize', 'xcorr_same',
    'xcorr_same_direct', 'build_icc', 'find_displacement', 'tile_size', 'fft_period',
    'cross_power_spectrum', 'upsampled_cc', 'upsampled_cc_window',
    'xcorr_refine', 'xcorr_refine_batch', 'find_displacement_batch',
    'QUAD_PINV_5X5', 'STATUS_OK', 'STATUS_EDGE', 'STATUS_NOMAX',
    'STATUS_OUTSIDE', 'STATUS_NONFINITE', 'primary_boxes', 'blot_affine4', 'blot_map4',
]

# per-item status codes shared with the HIP library (include/subpixal_hip.h)
STATUS_OK = 0        # quadratic vertex accepted
STATUS_EDGE = 1      # arg-max in row/column 0: integer peak (centroid.py:171-172)
STATUS_NOMAX = 2     # fitted quadric has no maximum: box centre (centroid.py:218-225)
STATUS_OUTSIDE = 3   # vertex outside the image: integer peak (centroid.py:230-236)
STATUS_NONFINITE = 6  # NaN/Inf input: the whole correlation is NaN, numpy.argmax returns 0 and
#                       find_peak the integer position (0, 0) (centroid.py:114, 171-172)


# --------------------------------------------------------------------------
# utils.py:144-161
# --------------------------------------------------------------------------
def py2round(x):
    """Round half away from zero (Python-2 ``round``), scalar or array.

    Restates ``subpixal/utils.py:144-161``.
    """
    if hasattr(x, '__iter__'):
        x = np.asarray(x)
        out = np.empty_like(x)
        pos = x >= 0.0
        out[pos] = np.floor(x[pos] + 0.5)
        out[~pos] = np.ceil(x[~pos] - 0.5)
        return out
    return np.floor(x + 0.5) if x >= 0.0 else np.ceil(x - 0.5)


# --------------------------------------------------------------------------
# centroid.py:239-253
# --------------------------------------------------------------------------
def _box_pars(par):
    """(wx, wy) from a scalar or a 2-sequence; centroid.py:239-253."""
    if hasattr(par, '__iter__'):
        if len(par) != 2:
            raise TypeError("Box specification must be either a scalar or "
                            "an iterable with two elements.")
        wx, wy = int(par[0]), int(par[1])
    else:
        wx = wy = int(par)
    if wx < 1 or wy < 1:
        raise ValueError("Box dimensions must be positive integer numbers.")
    return wx, wy


def _clip_box(c, w, n):
    """1-D fit/search box around index ``c``: centroid.py:140-143,165-168."""
    lo = max(0, c - w // 2)
    hi = min(n, lo + w)
    return lo, hi


# --------------------------------------------------------------------------
# centroid.py:18-236
# --------------------------------------------------------------------------
def find_peak(image_data, xmax=None, ymax=None, peak_fit_box=5,
              peak_search_box=None, mask=None, _status=None):
    """Sub-pixel peak of a 2-D array by a quadratic fit in a box around its
    (optionally box-restricted, optionally masked) arg-max.

    Restates ``subpixal/centroid.py:18-236`` including every early return.
    ``_status`` (a list) receives one STATUS_* code when given (oracle-only
    extension used to check the HIP kernel's status output).
    """
    def done(xy, st):
        if _status is not None:
            _status.append(st)
        return xy

    if (xmax is None) != (ymax is None):                       # :93-96
        raise ValueError("Both 'xmax' and 'ymax' must be either None or not "
                         "None")
    img = np.asarray(image_data, dtype=np.float64)             # :98
    ny, nx = img.shape

    if isinstance(peak_search_box, str):                       # :102-109
        if peak_search_box == 'fitbox':
            peak_search_box = peak_fit_box
        elif peak_search_box == 'off':
            peak_search_box = None
        elif peak_search_box == 'all':
            peak_search_box = img.shape

    expand = False
    if xmax is None:                                           # :111-127
        if mask is None:
            jmax, imax = np.unravel_index(np.argmax(img), img.shape)
        else:
            jj, ii = np.indices(img.shape)
            k = np.argmax(img[mask])
            imax, jmax = ii[mask][k], jj[mask][k]
        imax, jmax = int(imax), int(jmax)
        coord = (float(imax), float(jmax))
    else:                                                      # :129-156
        imax, jmax = int(py2round(xmax)), int(py2round(ymax))
        coord = (xmax, ymax)
        if peak_search_box is not None:
            sbx, sby = _box_pars(peak_search_box)
            sx1, sx2 = _clip_box(imax, sbx, nx)
            sy1, sy2 = _clip_box(jmax, sby, ny)
            if sx1 < sx2 and sy1 < sy2:
                sub = img[sy1:sy2, sx1:sx2]
                dj, di = np.unravel_index(np.argmax(sub), sub.shape)
                imax, jmax = int(di) + sx1, int(dj) + sy1
                coord = (float(imax), float(jmax))
            expand = (sbx != nx or sby != ny)

    def retry():
        return find_peak(img, None, None, (wx, wy), None, mask, _status)

    wx, wy = _box_pars(peak_fit_box)                           # :158
    if wx * wy < 6:                                            # :160-162
        return done(coord, STATUS_EDGE)

    x1, x2 = _clip_box(imax, wx, nx)                           # :165-168
    y1, y2 = _clip_box(jmax, wy, ny)
    if imax == x1 or imax == x2 or jmax == y1 or jmax == y2:   # :171-172
        return done((float(imax), float(jmax)), STATUS_EDGE)

    if x2 - x1 < wx:                                           # :175-179
        if x1 == 0:
            x2 = min(nx, x1 + wx)
        if x2 == nx:
            x1 = max(0, x2 - wx)
    if y2 - y1 < wy:                                           # :180-184
        if y1 == 0:
            y2 = min(ny, y1 + wy)
        if y2 == ny:
            y1 = max(0, y2 - wy)
    if (x2 - x1) * (y2 - y1) < 6:                              # :186-188
        return done(coord, STATUS_EDGE)

    gx, gy = np.meshgrid(np.arange(x1, x2), np.arange(y1, y2))  # :191-197
    gx = gx.ravel()
    gy = gy.ravel()
    basis = np.stack([np.ones_like(gx), gx, gy, gx * gy, gx * gx, gy * gy],
                     axis=1)
    vals = img[y1:y2, x1:x2].ravel()
    if mask is not None:                                       # :198-204
        good = np.asarray(mask)[y1:y2, x1:x2].ravel()
        basis, vals = basis[good], vals[good]
        if vals.size < 6:
            return done(coord, STATUS_EDGE)

    try:                                                       # :206-214
        coef = np.linalg.lstsq(basis, vals, rcond=None)[0]
    except np.linalg.LinAlgError:
        return retry() if expand else done(coord, STATUS_NOMAX)

    _, c10, c01, c11, c20, c02 = coef                          # :217-225
    det = 4 * c02 * c20 - c11**2
    if det <= 0 or ((c20 > 0.0 and c02 >= 0.0) or (c20 >= 0.0 and c02 > 0.0)):
        if expand:
            return retry()
        return done(((x1 + x2) / 2.0, (y1 + y2) / 2.0), STATUS_NOMAX)

    xm = (c01 * c11 - 2.0 * c02 * c10) / det                   # :227-228
    ym = (c10 * c11 - 2.0 * c01 * c20) / det
    if 0.0 < xm < (nx - 1.0) and 0.0 < ym < (ny - 1.0):        # :230-236
        return done((xm, ym), STATUS_OK)
    if expand:
        return retry()
    return done(coord, STATUS_OUTSIDE)


# --------------------------------------------------------------------------
# The hot-path configuration of find_peak as a constant operator
# (SURVEY.md 8 a-5): box-relative coordinates, c = pinv(V) d.
# --------------------------------------------------------------------------
def _quad_pinv(w=5):
    gx, gy = np.meshgrid(np.arange(w), np.arange(w))
    gx = gx.ravel().astype(np.float64)
    gy = gy.ravel().astype(np.float64)
    v = np.stack([np.ones_like(gx), gx, gy, gx * gy, gx * gx, gy * gy], axis=1)
    return np.linalg.pinv(v)


#: 6x25 pseudo-inverse of the quadratic design matrix on the (0..4)^2 box,
#: rows (1, x, y, xy, x^2, y^2); the HIP library embeds the same numbers.
QUAD_PINV_5X5 = _quad_pinv(5)


def find_peak_5x5_all(img):
    """``find_peak(img, peak_fit_box=5, peak_search_box='all')`` for images of
    at least 5x5, with the least squares done by the constant 6x25 operator in
    box-relative coordinates.  Returns ``(x, y, status)``.

    This is the arithmetic the HIP kernel implements; ``tests`` check it equals
    :func:`find_peak` (centroid.py:18-236 with cc.py:86's arguments).
    """
    img = np.asarray(img, dtype=np.float64)
    ny, nx = img.shape
    assert nx >= 5 and ny >= 5
    jmax, imax = np.unravel_index(np.argmax(img), img.shape)
    imax, jmax = int(imax), int(jmax)
    if imax == 0 or jmax == 0:
        return float(imax), float(jmax), STATUS_EDGE
    x1 = min(max(0, imax - 2), nx - 5)
    y1 = min(max(0, jmax - 2), ny - 5)
    c = QUAD_PINV_5X5 @ img[y1:y1 + 5, x1:x1 + 5].ravel()
    _, c10, c01, c11, c20, c02 = c
    det = 4 * c02 * c20 - c11 * c11
    if det <= 0 or ((c20 > 0.0 and c02 >= 0.0) or (c20 >= 0.0 and c02 > 0.0)):
        return x1 + 2.5, y1 + 2.5, STATUS_NOMAX
    xm = x1 + (c01 * c11 - 2.0 * c02 * c10) / det
    ym = y1 + (c10 * c11 - 2.0 * c01 * c20) / det
    if 0.0 < xm < nx - 1.0 and 0.0 < ym < ny - 1.0:
        return xm, ym, STATUS_OK
    return float(imax), float(jmax), STATUS_OUTSIDE


# --------------------------------------------------------------------------
# cc.py:131-156
# --------------------------------------------------------------------------
def normalize(ref, images, zero=False):
    """NCC / ZNCC pre-normalisation; restates ``subpixal/cc.py:131-156``.

    Statistics are pooled over the non-zero pixels of all ``images``; exact
    zeros stay zero; ``ref`` is centred/scaled on all its pixels using the
    statistics of the union mask.  Arithmetic stays in the input dtype.
    """
    masks = [im != 0 for im in images]
    pooled = np.hstack([im[m] for im, m in zip(images, masks)])
    mean = np.mean(pooled) if zero else 0.0
    std = np.std(pooled)
    out = []
    for im, m in zip(images, masks):
        im = im.copy()
        if zero:
            im[m] -= mean
        im[m] /= std
        out.append(im)
    union = np.zeros(masks[0].shape, dtype=bool)
    for m in masks:
        union |= m
    ref = ref.copy()
    if zero:
        ref -= np.mean(ref[union])
    ref /= np.std(ref[union])
    return ref, out


# --------------------------------------------------------------------------
# cc.py:114-117  (scipy.signal.fftconvolve(ref, im[::-1, ::-1], mode='same'))
# --------------------------------------------------------------------------
def xcorr_same(ref, im):
    """Linear cross-correlation, 'same' window, zero lag at (ny//2, nx//2):
    ``cc[j, i] = sum_{y,x} ref[y, x] * im[y - (j - ny//2), x - (i - nx//2)]``.

    Same library call as the reference (cc.py:114-117) so the float32 rounding
    is the reference's own.
    """
    return signal.fftconvolve(ref, im[::-1, ::-1], mode='same')


def xcorr_same_direct(ref, im):
    """The same quantity from its definition (O(n^4), small inputs only);
    pins the lag convention independently of any FFT."""
    ref = np.asarray(ref, dtype=np.float64)
    im = np.asarray(im, dtype=np.float64)
    ny, nx = ref.shape
    out = np.zeros((ny, nx))
    for j in range(ny):
        ly = j - ny // 2
        ys = slice(max(0, ly), min(ny, ny + ly))
        yi = slice(max(0, -ly), min(ny, ny - ly))
        for i in range(nx):
            lx = i - nx // 2
            xs = slice(max(0, lx), min(nx, nx + lx))
            xi = slice(max(0, -lx), min(nx, nx - lx))
            out[j, i] = np.sum(ref[ys, xs] * im[yi, xi])
    return out


# --------------------------------------------------------------------------
# cc.py:98-128
# --------------------------------------------------------------------------
def build_icc(ref, im00, im10, im01, im11, cc_type='NCC'):
    """Interlaced 2x cross-correlation image; restates ``cc.py:98-128``."""
    ims = [im00, im10, im01, im11]
    if any(tuple(im.shape) != tuple(ref.shape) for im in ims):
        raise ValueError("All cutouts must have same shape.")
    cc_type = cc_type.upper()
    if cc_type in ('NCC', 'ZNCC'):
        ref, ims = normalize(ref, ims, cc_type == 'ZNCC')
    ccs = tuple(xcorr_same(ref, im) for im in ims)
    ny, nx = ccs[0].shape
    icc = np.empty((2 * ny, 2 * nx), dtype=ccs[0].dtype)
    icc[0::2, 0::2] = ccs[0][::-1, ::-1]
    icc[0::2, 1::2] = ccs[1][::-1, ::-1]
    icc[1::2, 0::2] = ccs[2][::-1, ::-1]
    icc[1::2, 1::2] = ccs[3][::-1, ::-1]
    return icc, ccs


# --------------------------------------------------------------------------
# cc.py:21-95
# --------------------------------------------------------------------------
def find_displacement(ref_image, image00, image10, image01, image11,
                      cc_type='NCC', full_output=False, _status=None):
    """Displacement of ``image00`` relative to ``ref_image`` from the peak of
    the interlaced cross-correlation; restates ``cc.py:21-95``."""
    icc, ccs = build_icc(ref_image, image00, image10, image01, image11,
                         cc_type)
    if np.all(np.isfinite(icc)):
        xm, ym = find_peak(icc, peak_fit_box=5, peak_search_box='all',
                           _status=_status)
    else:
        # numpy.argmax ranks NaN above everything and returns the first one (centroid.py:114);
        # what the reference does next depends on where that is (edge rule, or lstsq failing on
        # NaN): the restatement stops at the integer position and flags the source.  An
        # overflowed (+inf) correlation is treated the same way.
        key = np.where(np.isnan(icc), np.inf, icc)
        jm, im = np.unravel_index(np.argmax(key), key.shape)
        xm, ym = float(im), float(jm)
        if _status is not None:
            _status.append(STATUS_NONFINITE)
    xc = (icc.shape[1] - 1) // 4
    yc = (icc.shape[0] - 1) // 4
    dx = 0.5 * xm - xc
    dy = 0.5 * ym - yc
    return (dx, dy, icc, ccs) if full_output else (dx, dy)


def find_displacement_batch(ref, im4, cc_type='NCC'):
    """``find_displacement`` over ``ref[N,ny,nx]``, ``im4[N,4,ny,nx]`` (order
    00,10,01,11).  Returns ``(dxdy[N,2] float64, status[N] int32)``."""
    n = ref.shape[0]
    out = np.empty((n, 2))
    status = np.empty(n, dtype=np.int32)
    for k in range(n):
        st = []
        out[k] = find_displacement(ref[k], im4[k, 0], im4[k, 1], im4[k, 2],
                                   im4[k, 3], cc_type=cc_type, _status=st)
        status[k] = st[-1]
    return out, status


# --------------------------------------------------------------------------
# Pair / upsample=U mode (not in the reference)
# --------------------------------------------------------------------------
def fft_period(ny, nx):
    """FFT period P of the pair mode's trigonometric interpolant (and of the HIP library's
    transforms) for an (ny, nx) cutout, n = max(ny, nx):

    * n <= 32: 64, n <= 64: 128 -- scipy's ``next_fast_len(2n-1)`` for n = 32, 64 (cc.py:114);
    * above: the smallest multiple of 64 for which the reference's 'same' window (the only lags
      cc.py:114-126 keeps) is free of circular aliasing: lag index ``l`` of the full linear
      correlation runs over ``[0, 2n-2]`` and the window over ``[(n-1)//2, (n-1)//2 + n - 1]``,
      so ``P > 2n - 2 - (n-1)//2`` suffices: 128 up to n = 85, 192 up to n = 128, 256 up to 170 ...

    At integer lags (upsample = 1, and the reference's 5-image mode) any alias-free period gives
    the SAME numbers as scipy's own choice; the period only enters the definition of the
    interpolant between the integer lags."""
    n = max(ny, nx)
    if n <= 32:
        return 64
    if n <= 85:
        return 128
    need = 2 * n - 2 - (n - 1) // 2 + 1
    return 64 * ((need + 63) // 64)


def tile_size(ny, nx):
    """Name of the kernel family of the HIP library that takes an (ny, nx) cutout: 32, 64,
    85 (the 64 tile's fold path, period 128) or 128 (period 192); larger cutouts use the
    general path with the period of :func:`fft_period`."""
    n = max(ny, nx)
    for t in (32, 64, 85, 128):
        if n <= t:
            return t
    return n


def cross_power_spectrum(ref, img, period):
    """``S[k] = FFT(ref)[k] * conj(FFT(img)[k])`` on the zero-padded
    ``period x period`` grid, float64.  Its inverse DFT at integer lag l is the
    linear cross-correlation ``sum ref[x] img[x - l]`` (|l| < period - n)."""
    r = np.fft.fft2(np.asarray(ref, np.float64), s=(period, period))
    i = np.fft.fft2(np.asarray(img, np.float64), s=(period, period))
    return r * np.conj(i)


def _pad_spectrum_1d(spec, period, up, axis):
    """Zero-pad a length-``period`` DFT axis to ``period*up`` keeping the
    signed frequencies and splitting the Nyquist bin between +-period/2."""
    spec = np.moveaxis(spec, axis, 0)
    h = period // 2
    out = np.zeros((period * up,) + spec.shape[1:], dtype=spec.dtype)
    out[:h] = spec[:h]
    out[-h + 1:] = spec[h + 1:] if h > 1 else spec[:0]
    out[h] = 0.5 * spec[h]
    out[-h] = out[-h] + 0.5 * spec[h]
    return np.moveaxis(out, 0, axis)


def upsampled_cc(ref, img, upsample):
    """The pair-mode fine cross-correlation image ``F[qy, qx]`` of shape
    ``(U*ny, U*nx)``: the real trigonometric interpolant (period P = fft_period(ny, nx) in both
    axes, Nyquist bin split symmetrically) of the zero-padded linear
    cross-correlation, sampled at lag ``(n - 1 - n//2) - q/U`` -- i.e. the
    reference's flipped 'same' window (cc.py:114-126) on a U-times finer grid.
    U=1 gives ``xcorr_same(ref, img)[::-1, ::-1]`` (up to rounding)."""
    ny, nx = ref.shape
    up = int(upsample)
    period = fft_period(ny, nx)
    spec = cross_power_spectrum(ref, img, period)
    if up > 1:
        spec = _pad_spectrum_1d(spec, period, up, 0)
        spec = _pad_spectrum_1d(spec, period, up, 1)
    fine = np.fft.ifft2(spec).real * (up * up)
    m = period * up
    iy = (up * (ny - 1 - ny // 2) - np.arange(up * ny)) % m
    ix = (up * (nx - 1 - nx // 2) - np.arange(up * nx)) % m
    return fine[np.ix_(iy, ix)]


def upsampled_cc_window(ref, img, upsample, qy, qx):
    """``upsampled_cc(ref, img, U)[np.ix_(qy, qx)]`` by a direct matrix DFT of
    the cross-power spectrum (no (U*P)^2 grid); used for large n*U."""
    ny, nx = ref.shape
    up = int(upsample)
    period = fft_period(ny, nx)
    spec = cross_power_spectrum(ref, img, period)
    k = np.fft.fftfreq(period, 1.0 / period)          # signed, -P/2 at index P/2
    wgt = np.ones(period)
    ty = (ny - 1 - ny // 2) - np.asarray(qy, np.float64) / up
    tx = (nx - 1 - nx // 2) - np.asarray(qx, np.float64) / up

    def basis(t):
        e = np.exp(2j * np.pi * np.outer(t, k) / period) * wgt
        # split Nyquist: replace e^{-i pi t} by cos(pi t)
        e[:, period // 2] = np.cos(np.pi * t)
        return e
    ey, ex = basis(ty), basis(tx)
    return (ey @ spec @ ex.T).real / (period * period)


def xcorr_refine(ref, img, upsample=1, cc_type='CC', _status=None,
                 full_grid=None):
    """Pair mode: shift of ``img`` relative to ``ref`` from the quadratic-fit
    peak (find_peak(., 5, 'all'), cc.py:86) of the U-times Fourier-upsampled
    cross-correlation; ``d = peak/U - (n-1)//2`` (cc.py:89-93 with 2 -> U).

    ``cc_type`` applies :func:`normalize` with the single image as the pool.
    Arithmetic is float64 throughout (the definition, not a timing path).
    Returns ``(dx, dy)``.
    """
    ny, nx = ref.shape
    up = int(upsample)
    cc_type = cc_type.upper()
    if cc_type in ('NCC', 'ZNCC'):
        ref, (img,) = normalize(ref, [img], cc_type == 'ZNCC')
    if full_grid is None:
        full_grid = (fft_period(ny, nx) * up) <= 2048
    st = []
    if not (np.all(np.isfinite(ref)) and np.all(np.isfinite(img))):
        # every lag of an FFT correlation is NaN then: numpy.argmax -> index 0 -> edge rule
        if _status is not None:
            _status.append(STATUS_NONFINITE)
        return 0.0 - (nx - 1) // 2, 0.0 - (ny - 1) // 2
    if full_grid:
        fine = upsampled_cc(ref, img, up)
        xm, ym = find_peak(fine, peak_fit_box=5, peak_search_box='all',
                           _status=st)
    else:
        # coarse arg-max, then the fine image only in a +-1.5 px window; the
        # fit box must not touch the window border (checked).
        coarse = upsampled_cc(ref, img, 1)
        jc, ic = np.unravel_index(np.argmax(coarse), coarse.shape)
        half = (3 * up) // 2 + 3
        qy = np.arange(max(0, jc * up - half), min(up * ny, jc * up + half + 1))
        qx = np.arange(max(0, ic * up - half), min(up * nx, ic * up + half + 1))
        win = upsampled_cc_window(ref, img, up, qy, qx)
        jw, iw = np.unravel_index(np.argmax(win), win.shape)
        jm, im = int(qy[jw]), int(qx[iw])
        inner_y = (jw >= 2 or qy[0] == 0) and (jw < len(qy) - 2 or qy[-1] == up * ny - 1)
        inner_x = (iw >= 2 or qx[0] == 0) and (iw < len(qx) - 2 or qx[-1] == up * nx - 1)
        if not (inner_y and inner_x):
            raise RuntimeError("fine peak left the refinement window; use "
                               "full_grid=True")
        # emulate find_peak on the virtual (U*ny, U*nx) image
        xm, ym, s = _peak_fit_virtual(win, int(qx[0]), int(qy[0]), up * nx,
                                      up * ny, im, jm)
        st.append(s)
    if _status is not None:
        _status.append(st[-1])
    dx = xm / up - (nx - 1) // 2
    dy = ym / up - (ny - 1) // 2
    return dx, dy


def _peak_fit_virtual(win, x0, y0, nx, ny, imax, jmax):
    """find_peak's box + fit step (centroid.py:158-236) for a peak already
    located at (imax, jmax) of a virtual nx x ny image of which ``win`` holds
    the part starting at (x0, y0)."""
    if imax == 0 or jmax == 0:
        return float(imax), float(jmax), STATUS_EDGE
    x1 = min(max(0, imax - 2), nx - 5)
    y1 = min(max(0, jmax - 2), ny - 5)
    box = win[y1 - y0:y1 - y0 + 5, x1 - x0:x1 - x0 + 5]
    assert box.shape == (5, 5)
    c = QUAD_PINV_5X5 @ box.ravel()
    _, c10, c01, c11, c20, c02 = c
    det = 4 * c02 * c20 - c11 * c11
    if det <= 0 or ((c20 > 0.0 and c02 >= 0.0) or (c20 >= 0.0 and c02 > 0.0)):
        return x1 + 2.5, y1 + 2.5, STATUS_NOMAX
    xm = x1 + (c01 * c11 - 2.0 * c02 * c10) / det
    ym = y1 + (c10 * c11 - 2.0 * c01 * c20) / det
    if 0.0 < xm < nx - 1.0 and 0.0 < ym < ny - 1.0:
        return xm, ym, STATUS_OK
    return float(imax), float(jmax), STATUS_OUTSIDE


def xcorr_refine_batch(ref, img, upsample=1, cc_type='CC', full_grid=None):
    """``xcorr_refine`` over ``ref[N,ny,nx]``, ``img[N,ny,nx]``.
    Returns ``(dxdy[N,2] float64, status[N] int32)``."""
    n = ref.shape[0]
    out = np.empty((n, 2))
    status = np.empty(n, dtype=np.int32)
    for k in range(n):
        st = []
        out[k] = xcorr_refine(ref[k], img[k], upsample, cc_type, _status=st,
                              full_grid=full_grid)
        status[k] = st[-1]
    return out, status


# --------------------------------------------------------------------------
# "Next" row: extraction boxes of the primary cutouts.  Restated from the
# reference source cutout.py:138-175 (that module cannot be imported here -- it needs
# astropy/stwcs at import time -- so this part is pinned by reading, not by execution).
# --------------------------------------------------------------------------
def primary_boxes(segmentation_image, ids=None, pad=1):
    """(kept_ids, boxes[(x0, y0, w, h)]) the way ``create_primary_cutouts`` derives them:
    one ``segmentation_image == sid`` scan per source (cutout.py:151-160), edge-touching
    sources skipped (162-167), padded (139, 169-173)."""
    seg = np.asarray(segmentation_image)
    ny, nx = seg.shape
    pad = int(np.ceil(pad)) if pad >= 0 else int(np.floor(pad))
    present = np.setdiff1d(np.unique(seg), [0])
    if ids is not None:
        present = np.intersect1d(np.asarray(ids), present)
    kept, boxes = [], []
    for sid in present:
        yy, xx = np.where(seg == sid)
        x1, x2, y1, y2 = xx.min(), xx.max(), yy.min(), yy.max()
        if x1 <= 0 or y1 <= 0 or x2 >= nx - 1 or y2 >= ny - 1:
            continue
        kept.append(sid)
        boxes.append((x1 - pad, y1 - pad, x2 - x1 + 1 + 2 * pad, y2 - y1 + 1 + 2 * pad))
    return (np.asarray(kept, dtype=np.int32),
            np.asarray(boxes, dtype=np.int32).reshape(len(boxes), 4))


# --------------------------------------------------------------------------
# Timing leg used by bench.py's cpu_baseline ("port"): the reference's own
# composition for one pair at U=1 -- fftconvolve 'same' + find_peak -- in the
# input dtype, exactly the per-pair work SURVEY.md section 6 timed.
# --------------------------------------------------------------------------
def pair_shift_u1(ref, img):
    """U=1 pair shift through the reference's own library calls (float32 FFT
    for float32 inputs): cc.py:114 + cc.py:86 + cc.py:89-93 for one image."""
    cc = xcorr_same(ref, img)[::-1, ::-1]
    xm, ym = find_peak(cc, peak_fit_box=5, peak_search_box='all')
    ny, nx = cc.shape
    return xm - (nx - 1) // 2, ym - (ny - 1) // 2


# ---------------------------------------------------------------------------
# Half-pixel dithered blots for affine maps (kernel: blot_affine4_kernel).
# The reference calls drizzlepac's tblot(interp='poly5') four times per source
# (align.py:664-676, blot.py:140-146); drizzlepac is absent from the reference tree, so
# this is the published quintic (the degree-5 polynomial through the six nearest samples
# per axis, separable -- what IRAF's bipoly5 / Everett's formula evaluates) written
# INDEPENDENTLY of the kernel as Lagrange weights in float64.  Parity with drizzlepac:
# UNPINNED.  Edge continuation v(-k) = 2 v(0) - v(k); outside the source -> 0.
# ---------------------------------------------------------------------------
def _lagrange6(s):
    """Weights of the samples at offsets -2..3 for a point at fractional offset s in [0, 1]."""
    nodes = np.arange(-2.0, 4.0)
    w = np.ones(6)
    for i in range(6):
        for j in range(6):
            if i != j:
                w[i] *= (s - nodes[j]) / (nodes[i] - nodes[j])
    return w


def _continued(tile, j, i):
    ny, nx = tile.shape

    def along_x(row, i):
        if i < 0:
            return 2.0 * tile[row, 0] - tile[row, -i]
        if i > nx - 1:
            return 2.0 * tile[row, nx - 1] - tile[row, 2 * (nx - 1) - i]
        return tile[row, i]

    if j < 0:
        return 2.0 * along_x(0, i) - along_x(-j, i)
    if j > ny - 1:
        return 2.0 * along_x(ny - 1, i) - along_x(2 * (ny - 1) - j, i)
    return along_x(j, i)


def _blot4(src, maps, ny, nx, gain=None):
    """``im4 [N, 4, ny, nx]`` float64 for per-source maps ``maps[b](xt, yt) -> (xs, ys)``."""
    src = np.asarray(src, dtype=np.float64)
    n, sny, snx = src.shape
    out = np.zeros((n, 4, ny, nx))
    for b in range(n):
        for q in range(4):
            ox, oy = 0.5 * (q & 1), 0.5 * ((q >> 1) & 1)
            for y in range(ny):
                for x in range(nx):
                    xs, ys = maps[b](x + ox, y + oy)
                    if not (0.0 <= xs <= snx - 1 and 0.0 <= ys <= sny - 1):
                        continue
                    ix, iy = int(np.floor(xs)), int(np.floor(ys))
                    wx, wy = _lagrange6(xs - ix), _lagrange6(ys - iy)
                    if ix >= 2 and ix + 3 < snx and iy >= 2 and iy + 3 < sny:
                        patch = src[b, iy - 2:iy + 4, ix - 2:ix + 4]
                    else:
                        patch = np.array([[_continued(src[b], iy - 2 + jj, ix - 2 + ii) for ii in range(6)]
                                          for jj in range(6)])
                    out[b, q, y, x] = wy @ patch @ wx
        if gain is not None:
            out[b] *= float(gain[b])
    return out


def blot_affine4(src, affine, ny, nx, gain=None):
    """``im4 [N, 4, ny, nx]`` float64: dithers 00, 10, 01, 11 <-> (ox, oy) in {0, 1/2}^2;
    ``imct.dx -= ox`` (align.py:668-676) puts cutout pixel x at image position x + blc - dx0 + ox
    (cutout.py:1138), i.e. the dither samples target position (x + ox, y + oy);
    target (x', y') -> source (a0 x' + a1 y' + a2, a3 x' + a4 y' + a5)."""
    affine = np.asarray(affine, dtype=np.float64)

    def mk(a):
        return lambda xt, yt: (a[0] * xt + a[1] * yt + a[2], a[3] * xt + a[4] * yt + a[5])
    return _blot4(src, [mk(a) for a in affine], ny, nx, gain)


def blot_map4(src, mappings, ny, nx, gain=None):
    """The same four blots through arbitrary per-source callables ``(xt, yt) -> (xs, ys)`` on the
    0-based un-dithered target grid (what BlotWCSMap does per pixel, blot.py:71-76): the float64
    reference the polynomial-map kernel ``spx_blot_poly4_f32`` is checked against."""
    return _blot4(src, list(mappings), ny, nx, gain)


# --------------------------------------------------------------------------
# the robust linear fit behind align.py:720-724
#   fit = linearfit.iter_linear_fit(xyim, xyref, wuv=weights, fitgeom=fitgeom, center=..., nclip=nclip,
#                                   sigma=sigma)
# tweakwcs (>= 0.4.2, setup.py:112) is un-vendored and absent here, and the reference holds no fixture
# for it: PARITY WITH tweakwcs IS UNPINNED.  This is an INDEPENDENT restatement of its documented
# behaviour -- weighted 'shift' / 'rscale' / 'general' fit of uv ~ F (xy - c) + c + t with iterative
# clipping of points whose residual exceeds sigma x the weighted RMS residual -- written with different
# algebra from subpixal_amd.align (one augmented lstsq; Procrustes/SVD for 'rscale'), so that
# tests/test_align_host.py checks the host code against something other than itself.
# --------------------------------------------------------------------------
def linear_fit_once(xy, uv, w, fitgeom):
    """Weighted least squares uv ~ F xy + t on points already referred to the centre.  (F, t)."""
    xy = np.asarray(xy, np.float64)
    uv = np.asarray(uv, np.float64)
    w = np.asarray(w, np.float64)
    if fitgeom == 'shift':
        return np.eye(2), np.average(uv - xy, axis=0, weights=w)
    if fitgeom == 'general':
        a = np.hstack([xy, np.ones((len(xy), 1))]) * np.sqrt(w)[:, None]
        sol = np.linalg.lstsq(a, uv * np.sqrt(w)[:, None], rcond=None)[0]       # (3, 2): rows x, y, 1
        return sol[:2].T, sol[2]
    if fitgeom == 'rscale':
        # weighted orthogonal Procrustes with scale (rotation + isotropic scale, no reflection)
        mx = np.average(xy, axis=0, weights=w)
        mu = np.average(uv, axis=0, weights=w)
        x, u = xy - mx, uv - mu
        h = (x * w[:, None]).T @ u
        um, s, vt = np.linalg.svd(h)
        d = np.sign(np.linalg.det(vt.T @ um.T))
        r = vt.T @ np.diag([1.0, d]) @ um.T
        scale = (s[0] + d * s[1]) / np.sum(w * np.sum(x * x, axis=1))
        f = scale * r
        return f, mu - f @ mx
    raise ValueError("Unsupported 'fitgeom'. Valid values are: 'shift', 'rscale', 'general'.")


def iter_linear_fit(xy, uv, wuv=None, fitgeom='general', center=None, nclip=3, sigma=3.0):
    """Returns ``(F, t, mask, n_iterations_that_clipped)``."""
    xy = np.asarray(xy, np.float64)
    uv = np.asarray(uv, np.float64)
    n = len(xy)
    w = np.ones(n) if wuv is None else np.asarray(wuv, np.float64)
    c = np.zeros(2) if center is None else np.asarray(center, np.float64)
    keep = w > 0
    need = {'shift': 1, 'rscale': 2, 'general': 3}[fitgeom]
    clipped = 0
    for it in range(int(nclip) + 1):
        f, t = linear_fit_once(xy[keep] - c, uv[keep] - c, w[keep], fitgeom)
        if it == nclip or sigma is None:
            break
        res = (uv - c) - ((xy - c) @ f.T + t)
        dist = np.hypot(res[:, 0], res[:, 1])
        rms = np.sqrt(np.average(dist[keep] ** 2, weights=w[keep]))
        new = keep & (dist <= sigma * rms)
        if new.sum() == keep.sum() or new.sum() < need:
            break
        keep = new
        clipped += 1
    return f, t, keep, clipped
