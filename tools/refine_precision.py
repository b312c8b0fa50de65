#!/usr/bin/env python
"""Where does pair mode's distance from the float64 definition come from: the refine stage's arithmetic
or the float32 transforms?  Same noise-free spots (SURVEY.md 8(d)'s parity set: sigma ~ U(4,6) px,
shifts ~ U(-3,3) px) through the kernel families that refine in float32 (64 tile, its fold path) and
the one that refines in float64 (period 192), against oracle.xcorr_refine_batch (float64 throughout),
per upsample factor.  Test infrastructure: the oracle is the checker here, nothing is timed.

    python tools/refine_precision.py [--count 128] [--budget 400]
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
ap.add_argument('--count', type=int, default=128)
ap.add_argument('--budget', type=float, default=400.0, help='seconds')
ap.add_argument('--refine64', default='float32', choices=['float32', 'float64'],
                help='what the loaded library\'s 64-tile kernel refines in (labels only: a -DSPX_REFINE64_F64=1 build is float64)')
a = ap.parse_args()

FAMILIES = [(64, '64 tile, %s refine' % a.refine64), (80, 'fold path, %s refine' % a.refine64),
            (96, 'period 192, float64 refine'), (128, 'period 192, float64 refine')]
UPS = [1, 2, 10, 20, 28, 40]
t0 = time.time()
print('pairs per cell: %d (float32 cutouts, noise-free); |kernel - float64 oracle| in px' % a.count)
print('%-4s %-28s %4s %10s %10s %10s' % ('n', 'family', 'U', 'median', '99 %', 'max'))
for n, fam in FAMILIES:
    ref, img, _ = datagen.pair_batch(20261005 + n, a.count, n, sigma_lo=4.0, sigma_hi=6.0)
    for up in UPS:
        if time.time() - t0 > a.budget:
            print('time budget reached'); sys.exit(0)
        got, st = spx.xcorr_refine_batch(ref, img, upsample=up, return_status=True)
        exp, est = orc.xcorr_refine_batch(ref, img, upsample=up)
        got = np.asarray(got); st = np.asarray(st)
        d = np.abs(got - exp).max(axis=1)
        print('%-4d %-28s %4d %10.2e %10.2e %10.2e   status equal: %s' % (
            n, fam, up, np.median(d), np.quantile(d, 0.99), d.max(), bool(np.array_equal(st, est))), flush=True)
