#!/usr/bin/env python
"""The general path (cutouts above 128 px) against the float64 definition, population by population: the
library refuses upsample >= 40 there because round 2's RANDOMIZED sweep (tools/sweep_parity.py: ragged shapes,
every cc type, two thirds of its cases with 0.3 % or 2 % noise) had bucket maxima of 1.1-1.2e-3 px at window
blocks 3-4.  Which inputs do that?  Needs a library built with -DSPX_MAX_UPSAMPLE_GENERAL=59 (SPX_HIP_LIB).
Test infrastructure: the oracle is the checker, nothing is timed.

    SPX_HIP_LIB=build_ab/lib_gen59.so python tools/general_precision.py [--count 12] [--budget 500]
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
ap.add_argument('--count', type=int, default=12)
ap.add_argument('--budget', type=float, default=500.0, help='seconds')
ap.add_argument('--shapes', default='160x160,200x200,200x140', help='NYxNX,... (any kernel family: the public API is called)')
ap.add_argument('--wide', default='11,25', help='sigma range of the wide spots, px')
ap.add_argument('--ups', default='20,39,43,59', help='upsample factors')
ap.add_argument('--refine', default='default', choices=['default', 'float64'], help='SPX_REFINE_* (33..85 px have two forms)')
a = ap.parse_args()

SHAPES = [tuple(int(v) for v in t.split('x')) for t in a.shapes.split(',')]
WLO, WHI = (float(v) for v in a.wide.split(','))
POPULATIONS = [('parity set: sigma 4..6 px, no noise', dict(sigma_lo=4.0, sigma_hi=6.0), 0.0),
               ('wide spots: sigma %g..%g px, no noise' % (WLO, WHI), dict(sigma_lo=WLO, sigma_hi=WHI), 0.0),
               ('robustness set: sigma 4..6 px, 1 % noise', dict(sigma_lo=4.0, sigma_hi=6.0), 0.01),
               ('wide spots, 2 % noise (the sweep\'s worst)', dict(sigma_lo=WLO, sigma_hi=WHI), 0.02)]
UPS = [int(v) for v in a.ups.split(',')]
t0 = time.time()
print('pairs per cell: %d (float32 cutouts), refine=%s; |kernel - float64 oracle| in px' % (a.count, a.refine))
print('%-44s %-9s %4s %10s %10s' % ('population', 'shape', 'U', 'median', 'max'))
for name, kw, noise in POPULATIONS:
    for ny, nx in SHAPES:
        tx, ty, sg, am = datagen.random_params(29, a.count, max(ny, nx), **kw)
        prs = [datagen.pair_set(ny, nx, tx[k], ty[k], sg[k], am[k], np.float32, noise_seed=1000 + k, noise_level=noise)
               for k in range(a.count)]
        ref = np.stack([p[0] for p in prs]); img = np.stack([p[1] for p in prs])
        for up in UPS:
            if time.time() - t0 > a.budget:
                print('time budget reached'); sys.exit(0)
            got, st = spx.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine=a.refine)
            exp, est = orc.xcorr_refine_batch(ref, img, upsample=up)
            d = np.abs(np.asarray(got) - exp).max(axis=1)
            print('%-44s %3dx%-5d %4d %10.2e %10.2e   status equal: %s' % (
                name, ny, nx, up, np.median(d), d.max(), bool(np.array_equal(np.asarray(st), est))), flush=True)
