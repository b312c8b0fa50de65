#!/usr/bin/env python
"""Soak check on real hardware: every kernel family, pair and reference mode, several launches of a
large batch must give bit-identical results (no atomics, fixed summation orders, barriers in the right
places), also under a permutation of the batch and across upsampling factors' window sizes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import datagen                                         # noqa: E402
import subpixal_amd as spx                             # noqa: E402
from subpixal_amd import synth                         # noqa: E402

REPS = int(os.environ.get('REPS', 6))
bad = 0
for n, count in ((32, 200000), (64, 100000), (80, 60000), (100, 30000), (128, 30000), (150, 1500)):
    gen = dict(sigma_lo=3.0, sigma_hi=4.0, max_shift=2.0) if n <= 32 else {}
    ref, img, truth = synth.gaussian_pairs(count, n, seed=31 + n, **gen)
    perm = torch.randperm(count, device=ref.device)
    for up in (1, 10, 20, 33):
        base, st0 = spx.xcorr_refine_batch(ref, img, upsample=up, cc_type='NCC', return_status=True)
        for _ in range(REPS - 1):
            d, st = spx.xcorr_refine_batch(ref, img, upsample=up, cc_type='NCC', return_status=True)
            if not (torch.equal(d, base) and torch.equal(st, st0)):
                bad += 1
                print('NON-DETERMINISTIC pair mode', n, up, int((d != base).any(1).sum()), flush=True)
        dp = spx.xcorr_refine_batch(ref[perm].contiguous(), img[perm].contiguous(), upsample=up, cc_type='NCC')
        if not torch.equal(dp, base[perm]):
            bad += 1
            print('PERMUTATION changes results', n, up, int((dp != base[perm]).any(1).sum()), flush=True)
    print('pair mode %3d px x %d: %d launches each at upsample 1/10/20/33 identical, err vs truth %.2e'
          % (n, count, REPS, float((base - truth).abs().max())), flush=True)
    r5, m4, _ = datagen.dither_batch(5, 64, n)
    reps = max(1, min(count, 20000 if n <= 128 else 640) // 64)
    r = torch.from_numpy(r5).cuda().repeat(reps, 1, 1).contiguous()
    m = torch.from_numpy(m4).cuda().repeat(reps, 1, 1, 1).contiguous()
    b5, i5 = spx.find_displacement_batch(r, m, cc_type='NCC', full_output=True)
    for _ in range(REPS - 1):
        d5, icc = spx.find_displacement_batch(r, m, cc_type='NCC', full_output=True)
        if not (torch.equal(d5, b5) and torch.equal(icc, i5)):
            bad += 1
            print('NON-DETERMINISTIC reference mode', n, flush=True)
    if not torch.equal(b5[:64], b5[-64:]):
        bad += 1
        print('reference mode: equal inputs, different outputs', n, flush=True)
    print('reference mode %3d px x %d: identical' % (n, r.shape[0]), flush=True)
print('soak: %d problems' % bad)
sys.exit(1 if bad else 0)
