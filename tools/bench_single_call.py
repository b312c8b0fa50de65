#!/usr/bin/env python
"""Latency of ONE reference-API call (the per-source call of align.py:682-685 after the
import swap of INTEGRATION.md A): numpy in, python floats out."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import datagen                                         # noqa: E402
import subpixal_amd                                    # noqa: E402
from subpixal_amd import centroid                      # noqa: E402

for n in (32, 64, 128):
    s = datagen.dither_set(n, n, 0.37, -0.81, 3.0)
    for full in (False, True):
        subpixal_amd.find_displacement(*s, cc_type='NCC', full_output=full)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            r = subpixal_amd.find_displacement(*s, cc_type='NCC', full_output=full)
        dt = (time.perf_counter() - t0) / reps
        print('find_displacement %3dx%-3d full_output=%-5s %7.1f us/call  (dx, dy) = (%.4f, %.4f)'
              % (n, n, full, dt * 1e6, r[0], r[1]))
img = np.random.default_rng(0).normal(size=(128, 128))
centroid.find_peak(img, peak_fit_box=5, peak_search_box='all')
t0 = time.perf_counter()
for _ in range(200):
    xy = centroid.find_peak(img, peak_fit_box=5, peak_search_box='all')
print('find_peak 128x128 %7.1f us/call' % ((time.perf_counter() - t0) / 200 * 1e6))
