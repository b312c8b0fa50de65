#!/usr/bin/env python
"""Randomized reference-mode sweep: cc.find_displacement for catalogs of sources with one cutout shape
each (3..140 px per side, all kernel families incl. the general path), CC/NCC/ZNCC, float32 and float64,
noise, thresholded (exactly-zero) pixels -- the GPU's one-launch-per-family path
(subpixal_amd.cc.find_displacement_var) against the oracle's per-source restatement of cc.py:21-95.

    python tools/sweep_disp5.py [--catalogs 40] [--sources 48] [--seed 1]
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
ap.add_argument('--catalogs', type=int, default=40)
ap.add_argument('--sources', type=int, default=48)
ap.add_argument('--seed', type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time()
worst = {}
bad = 0
total = 0
illcond = 0
for cat in range(a.catalogs):
    name = str(rng.choice(['CC', 'NCC', 'ZNCC']))
    dt = np.float64 if rng.random() < 0.3 else np.float32
    refs, ims = [], []
    for k in range(a.sources):
        big = rng.random() < 0.04
        ny = int(rng.integers(129, 141)) if big else int(rng.integers(3, 129))
        nx = int(rng.integers(3, 141)) if big else int(rng.integers(3, 129))
        small = min(ny, nx)
        smax = min(2.0, small / 6.0)
        t = datagen.dither_set(ny, nx, rng.uniform(-smax, smax), rng.uniform(-smax, smax),
                               max(0.8, small / rng.uniform(8, 14)), rng.uniform(0.5, 2.0), dt,
                               int(rng.integers(1, 1 << 30)) if rng.random() < 0.7 else 0,
                               float(rng.choice([0.003, 0.02])), int(rng.choice([0, 0, 1, 2])) if small >= 12 else 0)
        refs.append(t[0])
        ims.append(np.stack(t[1:]))
    d, iccs, st = spx.find_displacement_var(refs, ims, cc_type=name, full_output=True, return_status=True)
    for k in range(a.sources):
        s2 = []
        e = orc.find_displacement(refs[k], *ims[k], cc_type=name, _status=s2)
        err = float(np.max(np.abs(d[k] - np.array(e))))
        ny, nx = refs[k].shape
        fam = 32 if max(ny, nx) <= 32 else 64 if max(ny, nx) <= 64 else 85 if max(ny, nx) <= 85 else 128 if max(ny, nx) <= 128 else 200
        total += 1
        # north_star: 1e-3 px (wide spots on 100+ px cutouts have correlation peaks so flat that the float32
        # transforms limit the fit to a few 1e-4 px).  Tiny cutouts can have two arg-max candidates closer
        # than float32 resolves: those are skipped.
        if st[k] != s2[-1] or err > 1e-3:
            eicc = orc.build_icc(refs[k], *ims[k], cc_type=name)[0]
            flat = np.sort(eicc.ravel())[-2:]
            near_tie = (flat[1] - flat[0]) <= 3e-6 * abs(flat[1])
            if near_tie:
                continue                      # two candidates closer than float32 resolves
            # The reference accepts a fitted maximum anywhere inside the image (centroid.py:219-234): on a
            # nearly flat 5x5 box the quadratic's stationary point lands pixels away from the arg-max, OUTSIDE
            # the box it was fitted on, and moves by hundredths of a pixel when the values change in the 7th
            # digit.  Such sources are counted apart: if perturbations of the ORACLE's own float64 image at the
            # float32 transforms' accuracy (3e-7 relative, measured) move the ORACLE's answer by a sizeable
            # part of the disagreement, it is the fit's conditioning, not the kernel.
            if st[k] == s2[-1] and min(eicc.shape) >= 5:
                prng = np.random.default_rng(12345)
                base = np.array(orc.find_peak_5x5_all(eicc)[:2])
                jm, im_ = np.unravel_index(int(np.argmax(eicc)), eicc.shape)
                outside = max(abs(base[0] - im_), abs(base[1] - jm)) > 2.5
                wobble = max(float(np.max(np.abs(np.array(orc.find_peak_5x5_all(
                    eicc * (1.0 + 3e-7 * prng.standard_normal(eicc.shape)))[:2]) - base))) for _ in range(6)) / 2.0
                if wobble > 0.25 * err:
                    illcond += 1
                    print('ill-conditioned fit (skipped)', ny, nx, name, dt.__name__, 'err %.3g' % err,
                          'oracle moves %.3g px under 3e-7 relative noise; fitted maximum %s the 5x5 box'
                          % (wobble, 'outside' if outside else 'inside'), flush=True)
                    continue
            bad += 1
            print('MISMATCH', ny, nx, name, dt.__name__, err, st[k], s2[-1], flush=True)
        else:
            worst[fam] = max(worst.get(fam, 0.0), err)
print('%d sources in %d catalogs, %.0f s, %d mismatches, %d ill-conditioned fits skipped' % (total, a.catalogs, time.time() - t0, bad, illcond))
for fam in sorted(worst):
    print('family %3d: worst |d| = %.3g px' % (fam, worst[fam]))
sys.exit(1 if bad else 0)
