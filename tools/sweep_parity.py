#!/usr/bin/env python
"""Wider randomized parity sweep than tests/test_gpu_parity.py::test_randomized_shapes_sweep:
all five kernel families, every refinement-window size (upsample up to 59), the three cc types, noisy
cutouts; GPU against the oracle.  Prints the worst error per (tile, window blocks) bucket.

    python tools/sweep_parity.py [--trials 300] [--seed 1] [--budget 420]
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
ap.add_argument('--trials', type=int, default=300)
ap.add_argument('--seed', type=int, default=1)
ap.add_argument('--budget', type=float, default=420.0, help='seconds')
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time()
worst = {}
bad = 0
done = 0
for trial in range(a.trials):
    if time.time() - t0 > a.budget:
        break
    # kernel families: 32 tile, 64 tile, its fold path (65..85), period 192 (86..128), general path
    tile = int(rng.choice([32, 64, 85, 128, 200], p=[0.2, 0.3, 0.2, 0.2, 0.1]))
    lo = {32: 5, 64: 33, 85: 65, 128: 86, 200: 129}[tile]
    ny, nx = int(rng.integers(lo, tile + 1)), int(rng.integers(5, tile + 1))
    if rng.random() < 0.5:
        ny, nx = nx, ny
    if max(ny, nx) < lo:
        ny = int(rng.integers(lo, tile + 1))
    up = int(rng.choice([1, 2, 3, 4, 7, 10, 11, 16, 20, 26, 27, 33, 42, 43, 50, 59]))
    if tile == 200 and up > 39:
        up = int(rng.choice([33, 36, 39]))      # SPX_MAX_UPSAMPLE_GENERAL: above 128 px the library refuses finer grids
    name = str(rng.choice(['CC', 'NCC', 'ZNCC']))
    small = min(ny, nx)
    count = 2
    ref = np.empty((count, ny, nx), np.float32)
    img = np.empty_like(ref)
    for k in range(count):
        smax = min(2.5, small / 6.0)
        r, i = datagen.pair_set(ny, nx, rng.uniform(-smax, smax), rng.uniform(-smax, smax),
                                max(0.9, small / rng.uniform(8, 14)), rng.uniform(0.5, 2.0), np.float32,
                                noise_seed=int(rng.integers(1, 1 << 30)), noise_level=float(rng.choice([0.0, 0.003, 0.02])))
        ref[k], img[k] = r, i
    got, st = spx.xcorr_refine_batch(ref, img, upsample=up, cc_type=name, return_status=True)
    exp, est = orc.xcorr_refine_batch(ref, img, up, name)
    err = float(np.max(np.abs(got - exp)))
    key = (tile, (up + 5 + 15) // 16 if up > 1 else 0)
    worst[key] = max(worst.get(key, 0.0), err)
    done += 1
    # north_star: 1e-3 px.  (Round 1 needed 4e-3 on the 128 tile at upsample >= 20: float32 accumulation
    # of the fine window; above 85 px it accumulates in float64 now.)
    # (The general path (129+ px) reached 1.2e-3 px at upsample >= 40 in round 2 -- float32 transforms on peaks
    # that change by 1e-6 of their height across the fit box -- and refuses such grids since round 3:
    # SPX_MAX_UPSAMPLE_GENERAL = 39.  One limit everywhere.)
    limit = 1e-3
    if not np.array_equal(st, est) or err > limit:
        bad += 1
        print('MISMATCH', ny, nx, up, name, err, st, est, flush=True)
print('%d trials in %.0f s, %d mismatches' % (done, time.time() - t0, bad))
for key in sorted(worst):
    print('tile %3d, window blocks ~%d: worst |d| = %.3g px' % (key[0], key[1], worst[key]))
sys.exit(1 if bad else 0)
