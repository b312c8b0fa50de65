#!/usr/bin/env python
"""Where does pair mode's distance from the float64 definition come from: the refine stage's arithmetic
or the float32 transforms?  Same noise-free spots (SURVEY.md 8(d)'s parity set: sigma ~ U(4,6) px, 3..4 px on
the 32 tile, shifts ~ U(-3,3) px) through every kernel family up to 128 px, in both forms of the refine
where a family has two (refine='default' / 'float64' = SPX_REFINE_DEFAULT / SPX_REFINE_F64), against
oracle.xcorr_refine_batch (float64 throughout), per upsample factor.
Test infrastructure: the oracle is the checker here, nothing is timed.

    python tools/refine_precision.py [--count 48] [--budget 600]
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
ap.add_argument('--count', type=int, default=48)
ap.add_argument('--budget', type=float, default=600.0, help='seconds')
a = ap.parse_args()

# (cutout side, family, what each refine value means there)
FAMILIES = [(32, '32 tile', {'default': 'float32', 'float64': 'float32 (no other form)'}),
            (64, '64 tile', {'default': 'float32', 'float64': 'float64'}),
            (80, 'fold path', {'default': 'float32', 'float64': 'float64'}),
            (96, 'period 192', {'default': 'float64 (its only form)'}),
            (128, 'period 192', {'default': 'float64 (its only form)'})]
UPS = [1, 2, 10, 20, 28, 40]
t0 = time.time()
print('pairs per cell: %d (float32 cutouts, noise-free); |kernel - float64 oracle| in px' % a.count)
print('%-4s %-11s %-26s %4s %10s %10s %10s' % ('n', 'family', 'refine arithmetic', 'U', 'median', '99 %', 'max'))
for n, fam, forms in FAMILIES:
    ref, img, _ = datagen.pair_batch(20261005 + n, a.count, n)
    for up in UPS:
        if time.time() - t0 > a.budget:
            print('time budget reached'); sys.exit(0)
        exp, est = orc.xcorr_refine_batch(ref, img, upsample=up)
        for refine, what in forms.items():
            if refine == 'float64' and n <= 32 and up != 10:
                continue                                   # one line is enough to show it is the same kernel
            got, st = spx.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine=refine)
            d = np.abs(np.asarray(got) - exp).max(axis=1)
            print('%-4d %-11s %-26s %4d %10.2e %10.2e %10.2e   status equal: %s' % (
                n, fam, what, up, np.median(d), np.quantile(d, 0.99), d.max(),
                bool(np.array_equal(np.asarray(st), est))), flush=True)
