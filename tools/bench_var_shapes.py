#!/usr/bin/env python
"""A catalog's worth of cutouts with one shape per source (bounding box + padding: cutout.py:159-175):
`find_displacement_var` (one launch per kernel family) against the round-1 way (one launch per distinct
shape).  Host packing and PCIe included in both: numpy in, numpy out."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import datagen                                         # noqa: E402
from subpixal_amd import cc                            # noqa: E402

N = int(os.environ.get('N', 5000))
rng = np.random.default_rng(5)
base = {}
refs, ims = [], []
for k in range(N):
    ny, nx = int(rng.integers(18, 100)), int(rng.integers(18, 100))
    if (ny, nx) not in base:                           # one synthetic source per shape is enough here
        t = datagen.dither_set(ny, nx, 0.3, -0.4, max(1.5, min(ny, nx) / 12), 1.0, np.float32)
        base[(ny, nx)] = (t[0], np.stack(t[1:]))
    refs.append(base[(ny, nx)][0])
    ims.append(base[(ny, nx)][1])
print('%d sources, %d distinct shapes' % (N, len(base)))


def per_shape():
    groups = {}
    for k, r in enumerate(refs):
        groups.setdefault(r.shape, []).append(k)
    out = np.empty((N, 2))
    for shape, idx in groups.items():
        out[idx] = cc.find_displacement_batch(np.stack([refs[k] for k in idx]), np.stack([ims[k] for k in idx]),
                                              cc_type='NCC')
    return out


for name, fn in (('one launch per kernel family', lambda: cc.find_displacement_var(refs, ims, cc_type='NCC')),
                 ('one launch per shape', per_shape)):
    a = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-30s %8.1f ms  (%.0f sources/s)' % (name, 1e3 * dt, N / dt))
    if name.startswith('one launch per kernel'):
        keep = a
print('max |difference| between the two: %.2e px' % np.abs(keep - b).max())
