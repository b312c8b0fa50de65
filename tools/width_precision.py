#!/usr/bin/env python
"""Pair mode's distance from the float64 definition as a function of SPOT WIDTH (noise-free Gaussian spots in narrow
sigma bands, shifts ~ U(-3,3) px), per cutout size, upsample and refine form.  The wider the spot, the flatter the
correlation peak on the fine grid and the further float32 rounding in the transforms (and in a float32 refine) can move
the fitted vertex: this is what bounds the sizes / upsample factors over which "within 1e-3 px" can be promised.
Test infrastructure: the oracle is the checker, nothing is timed.

    python tools/width_precision.py [--count 32] [--shapes 64x64,85x85,128x128,200x200] [--ups 20,39,59]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import datagen                                         # noqa: E402
import subpixal_amd as spx                             # noqa: E402
from oracle import subpixal_oracle as orc              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--count', type=int, default=32)
ap.add_argument('--budget', type=float, default=600.0, help='seconds')
ap.add_argument('--shapes', default='64x64,85x85,128x128,200x200')
ap.add_argument('--ups', default='20,39,59')
ap.add_argument('--bands', default='4-6,6-8,8-11,11-15,15-20,20-25', help='sigma bands, px')
a = ap.parse_args()
SHAPES = [tuple(int(v) for v in t.split('x')) for t in a.shapes.split(',')]
UPS = [int(v) for v in a.ups.split(',')]
BANDS = [tuple(float(v) for v in b.split('-')) for b in a.bands.split(',')]
t0 = time.time()
print('pairs per cell: %d (float32 cutouts, noise-free); |kernel - float64 oracle| in px; ">1e-3" = pairs beyond the tolerance' % a.count)
print('%-9s %-9s %4s  %-8s %10s %10s %10s %6s' % ('shape', 'sigma', 'U', 'refine', 'median', '90 %', 'max', '>1e-3'))
for ny, nx in SHAPES:
    n = max(ny, nx)
    # 33..85 px: both forms requested EXPLICITLY (what 'default' means there depends on the upsample factor);
    # the other families have one form each
    forms = ['float32', 'float64'] if 32 < n <= 85 else ['default']
    for lo, hi in BANDS:
        tx, ty, sg, am = datagen.random_params(41, a.count, n, sigma_lo=lo, sigma_hi=hi)
        prs = [datagen.pair_set(ny, nx, tx[k], ty[k], sg[k], am[k]) for k in range(a.count)]
        ref = np.stack([p[0] for p in prs]); img = np.stack([p[1] for p in prs])
        for up in UPS:
            if time.time() - t0 > a.budget:
                print('time budget reached'); sys.exit(0)
            if n > 128 and up > 39:
                continue                                   # refused by the library above 128 px
            exp, est = orc.xcorr_refine_batch(ref, img, upsample=up)
            for refine in forms:
                got, st = spx.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine=refine)
                d = np.abs(np.asarray(got) - exp).max(axis=1)
                what = {'default': 'f32' if n <= 32 else 'f64', 'float32': 'f32', 'float64': 'f64'}[refine]
                print('%3dx%-5d %4g-%-4g %4d  %-8s %10.2e %10.2e %10.2e %6d%s' % (
                    ny, nx, lo, hi, up, what, np.median(d), np.quantile(d, 0.9), d.max(), int((d > 1e-3).sum()),
                    '' if np.array_equal(np.asarray(st), est) else '   STATUS DIFFERS'), flush=True)
