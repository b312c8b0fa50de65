"""Deterministic synthetic inputs shared by the golden generator, the parity
tests, smoke() and bench.py.  No reference code here.

``spot(ny, nx, x0, y0, s) = exp(-((x-x0)^2 + (y-y0)^2) / (2 s^2))`` on
``np.mgrid`` in float64, then cast.  The reference cutout has the spot at the
array centre ``c = (n-1)/2``; ``image00`` at ``c + (tx, ty)``; the half-pixel
dithers ``image10/01/11`` at ``x - 1/2``, ``y - 1/2``, both (the convention of
align.py:664-676: image10(x, y) = image00(x + 1/2, y)).
"""
import numpy as np

DTYPES = {0: np.float32, 1: np.float64}
CC_TYPES = {0: 'CC', 1: 'NCC', 2: 'ZNCC'}


def spot(ny, nx, x0, y0, sigma):
    y, x = np.mgrid[:ny, :nx].astype(np.float64)
    return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2.0 * sigma * sigma))


def dither_set(ny, nx, tx, ty, sigma, amp=1.0, dtype=np.float32,
               noise_seed=0, noise_level=0.0, zero_mode=0):
    """(ref, im00, im10, im01, im11) for one source.

    noise_seed > 0 adds independent N(0, noise_level*amp) noise to all five
    images (``default_rng(noise_seed)``).  zero_mode 1 sets pixels below
    1e-3*amp to exactly 0 (what a masked / thresholded cutout looks like to
    cc.py:135); zero_mode 2 zeroes a 5x7 block near the corner of every dither.
    """
    cx, cy = (nx - 1) / 2.0, (ny - 1) / 2.0
    ims = [
        amp * spot(ny, nx, cx, cy, sigma),
        amp * spot(ny, nx, cx + tx, cy + ty, sigma),
        amp * spot(ny, nx, cx + tx - 0.5, cy + ty, sigma),
        amp * spot(ny, nx, cx + tx, cy + ty - 0.5, sigma),
        amp * spot(ny, nx, cx + tx - 0.5, cy + ty - 0.5, sigma),
    ]
    if noise_seed:
        rng = np.random.default_rng(int(noise_seed))
        noise = rng.standard_normal((5, ny, nx)) * (noise_level * amp)
        ims = [im + nz for im, nz in zip(ims, noise)]
    if zero_mode == 1:
        ims = [np.where(np.abs(im) < 1e-3 * amp, 0.0, im) for im in ims]
    elif zero_mode == 2:
        for im in ims[1:]:
            im[1:6, 2:9] = 0.0
    return tuple(np.ascontiguousarray(im.astype(dtype)) for im in ims)


def pair_set(ny, nx, tx, ty, sigma, amp=1.0, dtype=np.float32,
             noise_seed=0, noise_level=0.0):
    """(ref, img) -- the first two images of :func:`dither_set`."""
    s = dither_set(ny, nx, tx, ty, sigma, amp, dtype, noise_seed, noise_level)
    return s[0], s[1]


def random_params(seed, count, n, sigma_lo=None, sigma_hi=None, max_shift=3.0):
    """BASELINE parity-set parameters: tx,ty ~ U(-3,3),
    sigma ~ U(4,6) (n=32: U(3,4)), amplitude ~ U(0.5,2)."""
    if sigma_lo is None:
        sigma_lo, sigma_hi = (3.0, 4.0) if n <= 32 else (4.0, 6.0)
    rng = np.random.default_rng([20261003, int(seed), int(n)])
    tx = rng.uniform(-max_shift, max_shift, count)
    ty = rng.uniform(-max_shift, max_shift, count)
    sigma = rng.uniform(sigma_lo, sigma_hi, count)
    amp = rng.uniform(0.5, 2.0, count)
    return tx, ty, sigma, amp


def pair_batch(seed, count, n, dtype=np.float32, **kw):
    """``ref[N,n,n], img[N,n,n], truth[N,2]`` for the parity set."""
    tx, ty, sigma, amp = random_params(seed, count, n, **kw)
    ref = np.empty((count, n, n), dtype=dtype)
    img = np.empty((count, n, n), dtype=dtype)
    for k in range(count):
        ref[k], img[k] = pair_set(n, n, tx[k], ty[k], sigma[k], amp[k], dtype)
    return ref, img, np.stack([tx, ty], axis=1)


def dither_batch(seed, count, n, dtype=np.float32, **kw):
    """``ref[N,n,n], im4[N,4,n,n], truth[N,2]`` for the 5-image mode."""
    tx, ty, sigma, amp = random_params(seed, count, n, **kw)
    ref = np.empty((count, n, n), dtype=dtype)
    im4 = np.empty((count, 4, n, n), dtype=dtype)
    for k in range(count):
        s = dither_set(n, n, tx[k], ty[k], sigma[k], amp[k], dtype)
        ref[k] = s[0]
        im4[k] = np.stack(s[1:])
    return ref, im4, np.stack([tx, ty], axis=1)
