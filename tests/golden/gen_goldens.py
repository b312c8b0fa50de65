#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container: it imports ``/root/reference/subpixal/cc.py``
and ``centroid.py`` IN PLACE through a stub package (the real package init needs
a generated version.py plus astropy/drizzlepac/stwcs/tweakwcs, none of which the
hot path touches).  Nothing of the reference is copied:
the committed fixtures hold parameters/small inputs and the reference's outputs.

    python tests/golden/gen_goldens.py        # rewrites tests/golden/*.npz

Environment the committed fixtures were made with: Python 3.10.12,
numpy 2.2.6, scipy 1.15.3 (the versions of this image).
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402

REF_ROOT = '/root/reference/subpixal'


def import_reference():
    sys.dont_write_bytecode = True
    pkg = types.ModuleType('subpixal')
    pkg.__path__ = [REF_ROOT]
    pkg.__version__ = 'golden'
    pkg.__version_date__ = 'golden'
    sys.modules['subpixal'] = pkg
    for name in ('astropy', 'astropy.io', 'astropy.io.fits'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['astropy'].io = sys.modules['astropy.io']
    sys.modules['astropy.io'].fits = sys.modules['astropy.io.fits']
    from subpixal import cc, centroid, utils
    return cc, centroid, utils


def gen_find_displacement(cc):
    """5-image mode: parameters + reference (dx, dy) and a few icc probes."""
    rows = []
    rng = np.random.default_rng(20261003)
    plans = [  # (ny, nx, count)
        (32, 32, 12), (64, 64, 12), (128, 128, 4), (20, 31, 6), (33, 33, 6),
        (48, 64, 4), (64, 40, 4), (8, 8, 3), (5, 6, 3),
    ]
    for ny, nx, count in plans:
        small = min(ny, nx)
        for k in range(count):
            smax = min(3.0, small / 8.0)
            tx, ty = rng.uniform(-smax, smax, 2)
            if small >= 64:
                sigma = rng.uniform(4.0, 6.0)
            elif small >= 20:
                sigma = rng.uniform(2.0, 4.0)
            else:
                sigma = rng.uniform(0.8, 1.5)
            amp = rng.uniform(0.5, 2.0)
            variant = k % 4       # 0,1 clean; 2 noisy; 3 zeros
            noise_seed = int(rng.integers(1, 2**31)) if variant == 2 else 0
            noise_level = 0.01 if variant == 2 else 0.0
            zero_mode = (1 + (k // 4) % 2) if variant == 3 else 0
            for dt in (0, 1):
                for ct in (0, 1, 2):
                    rows.append(dict(ny=ny, nx=nx, tx=tx, ty=ty, sigma=sigma,
                                     amp=amp, noise_seed=noise_seed,
                                     noise_level=noise_level,
                                     zero_mode=zero_mode, dtype=dt, cc_type=ct))
    out = {k: np.array([r[k] for r in rows]) for k in rows[0]}
    n = len(rows)
    dx = np.empty(n)
    dy = np.empty(n)
    icc_sum = np.empty(n)
    icc_max = np.empty(n)
    icc_argmax = np.empty(n, dtype=np.int64)
    in_sum = np.empty(n)
    for i, r in enumerate(rows):
        ims = datagen.dither_set(r['ny'], r['nx'], r['tx'], r['ty'], r['sigma'],
                                 r['amp'], datagen.DTYPES[r['dtype']],
                                 r['noise_seed'], r['noise_level'],
                                 r['zero_mode'])
        d = cc.find_displacement(*ims, cc_type=datagen.CC_TYPES[r['cc_type']],
                                 full_output=True)
        dx[i], dy[i] = d[0], d[1]
        icc = d[2]
        icc_sum[i] = np.sum(icc, dtype=np.float64)
        icc_max[i] = np.max(icc)
        icc_argmax[i] = int(np.argmax(icc))
        in_sum[i] = sum(float(np.sum(im, dtype=np.float64)) for im in ims)
    out.update(dx=dx, dy=dy, icc_sum=icc_sum, icc_max=icc_max,
               icc_argmax=icc_argmax, in_sum=in_sum)
    # two complete icc images (small) for full_output parity
    for tag, (ny, nx, ct) in {'a': (20, 31, 2), 'b': (32, 32, 0)}.items():
        ims = datagen.dither_set(ny, nx, 0.37, -0.81, 2.0, 1.3, np.float32)
        d = cc.find_displacement(*ims, cc_type=datagen.CC_TYPES[ct],
                                 full_output=True)
        out['full_%s_shape' % tag] = np.array([ny, nx, ct])
        out['full_%s_icc' % tag] = d[2]
        out['full_%s_ccs' % tag] = np.stack(d[3])
        out['full_%s_dxdy' % tag] = np.array(d[:2])
    np.savez_compressed(os.path.join(HERE, 'find_displacement.npz'), **out)
    return n


def gen_pair_u1(cc_mod, centroid):
    """Pair composition at U=1 with the reference's own calls:
    fftconvolve(ref, img[::-1,::-1],'same')[::-1,::-1] -> find_peak(.,5,'all')
    -> minus (n-1)//2 (cc.py:114, :86, :89-93 applied to one image)."""
    from scipy import signal
    rows = []
    for n, count in ((32, 24), (64, 24), (128, 6), (33, 6)):
        tx, ty, sigma, amp = datagen.random_params(1, count, n)
        for k in range(count):
            for dt in (0, 1):
                rows.append(dict(n=n, tx=tx[k], ty=ty[k], sigma=sigma[k],
                                 amp=amp[k], dtype=dt))
    out = {k: np.array([r[k] for r in rows]) for k in rows[0]}
    dx = np.empty(len(rows))
    dy = np.empty(len(rows))
    for i, r in enumerate(rows):
        ref, img = datagen.pair_set(r['n'], r['n'], r['tx'], r['ty'],
                                    r['sigma'], r['amp'],
                                    datagen.DTYPES[r['dtype']])
        c = signal.fftconvolve(ref, img[::-1, ::-1], mode='same')[::-1, ::-1]
        xm, ym = centroid.find_peak(c, peak_fit_box=5, peak_search_box='all')
        dx[i] = xm - (r['n'] - 1) // 2
        dy[i] = ym - (r['n'] - 1) // 2
    out.update(dx=dx, dy=dy)
    np.savez_compressed(os.path.join(HERE, 'pair_u1.npz'), **out)
    return len(rows)


def gen_bench_parity(cc):
    """The BASELINE parity set (sigma >= 4 at n >= 64): reference 5-image
    result on analytic dithers, to which the pair mode at U=2/U=10 is compared
    (north_star: within 1e-3 px)."""
    out = {}
    for n, count in ((32, 48), (64, 64), (128, 8)):
        tx, ty, sigma, amp = datagen.random_params(7, count, n)
        d = np.empty((count, 2))
        for k in range(count):
            ims = datagen.dither_set(n, n, tx[k], ty[k], sigma[k], amp[k],
                                     np.float32)
            d[k] = cc.find_displacement(*ims, cc_type='CC')
        out['n%d_dxdy' % n] = d
        out['n%d_params' % n] = np.stack([tx, ty, sigma, amp], axis=1)
    np.savez_compressed(os.path.join(HERE, 'bench_parity.npz'), **out)
    return sum(v.shape[0] for k, v in out.items() if k.endswith('dxdy'))


def _blob(rng, ny, nx, sharp=False):
    y, x = np.mgrid[:ny, :nx].astype(np.float64)
    x0 = rng.uniform(0, nx - 1)
    y0 = rng.uniform(0, ny - 1)
    s = rng.uniform(0.7, 1.2) if sharp else rng.uniform(1.2, 3.0)
    img = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * s * s))
    return img + 0.01 * rng.standard_normal((ny, nx))


def gen_find_peak(centroid, utils):
    rng = np.random.default_rng(424242)
    cases = []   # (img, mask or None, kwargs)

    degenerate = set()

    def add(img, mask=None, _degenerate=False, **kw):
        # _degenerate: the fitted curvature is exactly zero, so the sign of the
        # reference's det (centroid.py:218) is decided by lstsq rounding noise
        if _degenerate:
            degenerate.add(len(cases))
        cases.append((np.asarray(img, dtype=np.float64), mask, kw))

    shapes = [(10, 10), (12, 14), (7, 9), (5, 5), (6, 5), (16, 16), (9, 21)]
    for (ny, nx) in shapes:
        for rep in range(4):
            img = _blob(rng, ny, nx)
            add(img)                                       # defaults
            add(img, peak_search_box='all')
            add(img, peak_fit_box=3, peak_search_box='all')
            add(img, peak_fit_box=7)
            add(img, peak_fit_box=(3, 5))
            add(img, peak_fit_box=(5, 3), peak_search_box='off')
            jm, im = np.unravel_index(np.argmax(img), img.shape)
            gx = float(im) + rng.uniform(-2.5, 2.5)
            gy = float(jm) + rng.uniform(-2.5, 2.5)
            gx = min(max(gx, 0.0), nx - 1.0)
            gy = min(max(gy, 0.0), ny - 1.0)
            add(img, xmax=gx, ymax=gy)
            add(img, xmax=gx, ymax=gy, peak_search_box='fitbox')
            add(img, xmax=gx, ymax=gy, peak_search_box='all')
            add(img, xmax=gx, ymax=gy, peak_search_box=3)
            add(img, xmax=gx, ymax=gy, peak_search_box=(7, 3))
            add(img, xmax=float(im) + 0.5, ymax=float(jm) - 0.5,
                peak_search_box='off')
            good = rng.uniform(size=img.shape) > 0.15
            add(img, mask=good)
            add(img, mask=good, peak_fit_box=3)
            add(img, mask=good, xmax=gx, ymax=gy, peak_search_box=5)
            notmax = np.ones(img.shape, dtype=bool)
            notmax[jm, im] = False
            add(img, mask=notmax)
    # small / degenerate fit boxes
    img = _blob(rng, 8, 8)
    add(img, peak_fit_box=2)
    add(img, peak_fit_box=1)
    add(img, peak_fit_box=(2, 3))
    add(img, peak_fit_box=(1, 9))
    add(_blob(rng, 3, 3))
    add(_blob(rng, 4, 2), peak_fit_box=3)
    add(_blob(rng, 2, 7), peak_fit_box=5)
    # peaks on edges / corners
    for (j, i) in [(0, 4), (9, 4), (4, 0), (4, 9), (0, 0), (9, 9), (1, 4),
                   (8, 4), (4, 1), (4, 8), (1, 1), (8, 8)]:
        y, x = np.mgrid[:10, :10].astype(np.float64)
        img = np.exp(-((x - i - 0.2) ** 2 + (y - j + 0.1) ** 2) / 4.0)
        add(img)
        add(img, peak_search_box='all', peak_fit_box=5)
        add(img, xmax=float(i), ymax=float(j), peak_search_box=3)
    # flat, saddle, bowl, ramp
    y, x = np.mgrid[:11, :13].astype(np.float64)
    add(np.zeros((10, 10)))
    add(np.ones((6, 7)))
    add((x - 6) ** 2 - (y - 5) ** 2)
    add((x - 6) ** 2 + (y - 5) ** 2)
    add(x + 2 * y, _degenerate=True)
    add(-((x - 6.3) ** 2) - 0 * y, _degenerate=True)     # ridge: c02 == 0
    add(-(x - 6.3) ** 2 - (y - 4.6) ** 2)                # exact paraboloid
    add(-(x - 6.3) ** 2 - 3 * (y - 4.6) ** 2 + 0.5 * (x - 6.3) * (y - 4.6))
    # auto_expand_search recursion: guess far from the true peak, small box
    for rep in range(6):
        img = _blob(rng, 14, 14, sharp=True)
        add(img, xmax=rng.uniform(0, 13), ymax=rng.uniform(0, 13),
            peak_search_box=3)
        add(img, xmax=rng.uniform(0, 13), ymax=rng.uniform(0, 13),
            peak_search_box=(3, 5), peak_fit_box=3)
        add(-img, xmax=rng.uniform(2, 11), ymax=rng.uniform(2, 11),
            peak_search_box=3)
    # the vertex-outside-image branch: steep one-sided slope near the border
    y, x = np.mgrid[:8, :8].astype(np.float64)
    add(np.exp(0.9 * x) - 0.02 * (y - 3.5) ** 2)
    add(np.exp(0.9 * y) - 0.02 * (x - 3.5) ** 2)

    arrays = {}
    meta = []
    for k, (img, mask, kw) in enumerate(cases):
        arrays['img_%03d' % k] = img
        if mask is not None:
            arrays['mask_%03d' % k] = mask
        res = centroid.find_peak(img, mask=mask, **kw)
        jkw = {a: (list(v) if isinstance(v, tuple) else v) for a, v in kw.items()}
        meta.append(dict(kwargs=jkw, has_mask=mask is not None, degenerate=k in degenerate,
                         expected=[float(res[0]), float(res[1])]))
    # error conventions
    errors = []
    img = _blob(rng, 8, 8)
    arrays['img_err'] = img
    for kw in (dict(xmax=1.0), dict(ymax=2.0), dict(peak_fit_box=(1, 2, 3)),
               dict(peak_fit_box=0), dict(peak_fit_box=(3, -1)),
               dict(xmax=3.0, ymax=3.0, peak_search_box=(1, 2, 3)),
               dict(xmax=3.0, ymax=3.0, peak_search_box=0)):
        try:
            centroid.find_peak(img, **kw)
            name = None
        except Exception as e:       # noqa: BLE001 - record whatever it raises
            name = type(e).__name__
        jkw = {a: (list(v) if isinstance(v, tuple) else v) for a, v in kw.items()}
        errors.append(dict(kwargs=jkw, raises=name))
    # py2round KATs
    xs = np.array([-2.5, -1.5, -0.5, -0.49, 0.0, 0.49, 0.5, 1.5, 2.5, 3.49999,
                   -7.500001, 1e6 + 0.5])
    arrays['py2round_x'] = xs
    arrays['py2round_scalar'] = np.array([float(utils.py2round(float(v)))
                                          for v in xs])
    arrays['py2round_array'] = utils.py2round(xs)
    arrays['meta_json'] = np.array(json.dumps(dict(cases=meta, errors=errors)))
    np.savez_compressed(os.path.join(HERE, 'find_peak.npz'), **arrays)
    return len(cases)


def main():
    cc, centroid, utils = import_reference()
    print('find_displacement cases:', gen_find_displacement(cc))
    print('pair_u1 cases:', gen_pair_u1(cc, centroid))
    print('bench_parity cases:', gen_bench_parity(cc))
    print('find_peak cases:', gen_find_peak(centroid, utils))


if __name__ == '__main__':
    main()
