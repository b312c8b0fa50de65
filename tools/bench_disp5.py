#!/usr/bin/env python
"""Throughput of the reference mode (cc.find_displacement for a batch): displacements/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import subpixal_amd, datagen
N = int(os.environ.get("N", 20000))
SIZES = [int(v) for v in os.environ.get('SIZES', '32,64,80,96,128,160').split(',')]
TYPES = os.environ.get('CC_TYPES', 'CC,NCC,ZNCC').split(',')
for n in SIZES:
    ref, im4, truth = datagen.dither_batch(3, 64, n)
    reps = (N if n <= 128 else N // 8) // 64
    r = torch.from_numpy(ref).cuda().repeat(reps, 1, 1).contiguous()
    m = torch.from_numpy(im4).cuda().repeat(reps, 1, 1, 1).contiguous()
    for cc in TYPES:
        subpixal_amd.find_displacement_batch(r, m, cc_type=cc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            d = subpixal_amd.find_displacement_batch(r, m, cc_type=cc)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print('n=%3d %-4s %9.0f displacements/s (%9.0f cross-correlations/s)' % (n, cc, r.shape[0] / dt, 4 * r.shape[0] / dt))
