#!/usr/bin/env python
"""Kernel cases for the sanitizer builds of the CPU harness (tools/run_emu_asan.sh,
tools/run_emu_tsan.sh): every tile and every refinement-window size, with several pairs per
workgroup so that the waves run on across pair boundaries (the pair kernel has no barrier at
the end of a pair).  Results are checked against the oracle as well."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import datagen                                         # noqa: E402
import emu                                             # noqa: E402
from oracle import subpixal_oracle as orc              # noqa: E402

quick = '--quick' in sys.argv          # thread sanitizer: 10-20x slower, fewer window sizes
cases = [(64, 6, 2, (1, 10) if quick else (1, 2, 10, 16, 20, 30, 43, 59)),
         (50, 2, 1, () if quick else (27,)),
         (63, 2, 1, () if quick else (10,)),                      # ragged rows: shifted chunk loads
         (32, 10, 1, (10,) if quick else (1, 10, 20, 43)),
         (21, 5, 1, () if quick else (59,)),
         (70, 3, 1, (10,) if quick else (1, 10, 43)),             # 64 tile, fold path
         (85, 2, 1, () if quick else (2, 27)),
         (65, 1, 1, () if quick else (59,)),
         (96, 2, 1, (10,) if quick else (1, 2, 10, 27)),          # period 192
         (128, 2, 1, (2,) if quick else (1, 11, 20, 43)),
         (97, 1, 1, () if quick else (59,)),
         (150, 1, 1, () if quick else (1, 10)),                   # general path
         (131, 1, 1, () if quick else (30,))]
for n, count, grid, ups in cases:
    ref, img, truth = datagen.pair_batch(7, count, n)
    for up in ups:
        emu.set_grid(grid)
        try:
            got, st = emu.pair(ref, img, up)
        finally:
            emu.set_grid(0)
        exp, est = orc.xcorr_refine_batch(ref, img, up)
        err = float(np.abs(got - exp).max())
        print('pair %3dx%-3d U=%-2d max |d| %.2e' % (n, n, up, err), flush=True)
        assert np.array_equal(st, est) and err < 2e-3
# non-finite pixels and float64 inputs on every family
for n in (20, 64, 80, 100) if not quick else (64,):
    ref, img, truth = datagen.pair_batch(9, 2, n)
    img = img.copy()
    img[0, 3, 4] = np.nan
    for up in (1, 10):
        got, st = emu.pair(ref, img, up)
        assert st[0] == 6 and st[1] == 0
    got, st = emu.pair(ref.astype(np.float64), datagen.pair_batch(9, 2, n)[1].astype(np.float64), 10, 2)
    exp, est = orc.xcorr_refine_batch(ref.astype(np.float64), datagen.pair_batch(9, 2, n)[1].astype(np.float64), 10, 'ZNCC')
    print('nan/f64 %3d ok, f64 ZNCC max |d| %.2e' % (n, float(np.abs(got - exp).max())), flush=True)
    assert np.array_equal(st, est) and np.abs(got - exp).max() < 1e-3
# cutouts narrower than a 4-pixel load chunk (reference mode goes down to 3 px)
for shape in ((50, 3), (3, 50), (100, 3), (80, 3), (20, 3), (130, 3)):
    t = datagen.dither_set(shape[0], shape[1], 0.2, -0.3, 0.9, 1.0, np.float32)
    d, st, icc = emu.disp5(t[0][None], np.stack(t[1:])[None], 1)
    e = orc.find_displacement(*t, cc_type='NCC')
    assert np.abs(d[0] - np.array(e)).max() < 1e-4, shape
print('narrow cutouts OK', flush=True)
for n in (24, 48) if quick else (24, 48, 77, 100, 140):
    r, m4, t = datagen.dither_batch(3, 2, n)
    d, icc, st = emu.disp5(r, m4, 1)
    e, est = orc.find_displacement_batch(r, m4, 'NCC')
    print('find_displacement %3dx%-3d max |d| %.2e' % (n, n, float(np.abs(d - e).max())), flush=True)
    assert np.abs(d - e).max() < 1e-4
# round 3: the eight-wave pair kernel (spx_kernels8.h) and the five-transform reference-mode kernel
# (spx_kernels5.h), several items per workgroup
ref, img, truth = datagen.pair_batch(11, 5, 64)
for up in (1, 10) if quick else (1, 10, 20, 43):
    emu.set_grid(2)
    try:
        got, st = emu.pair(ref, img, up, tile=648)
    finally:
        emu.set_grid(0)
    exp, est = orc.xcorr_refine_batch(ref, img, up)
    print('eight-wave pair 64x64 U=%-2d max |d| %.2e' % (up, float(np.abs(got - exp).max())), flush=True)
    assert np.array_equal(st, est) and np.abs(got - exp).max() < 1e-3
emu.set_disp5_packed(2)
try:
    for n in (48,) if quick else (48, 64, 77, 85):
        r, m4, t = datagen.dither_batch(5, 3, n)
        for cc, name in ((0, 'CC'), (2, 'ZNCC')):
            d, st, icc = emu.disp5(r, m4, cc)
            e, est = orc.find_displacement_batch(r, m4, name)
            print('five-transform find_displacement %3dx%-3d %s max |d| %.2e' % (n, n, name, float(np.abs(d - e).max())), flush=True)
            assert np.array_equal(st, est) and np.abs(d - e).max() < 1e-4
finally:
    emu.set_disp5_packed(1)
# round 3: the float64 form of the 64 tile's refine (spx_kernels.h RefineF64: float64 tables, result rows lk + 4 r,
# block-at-a-time stage 2 from three window blocks on), ragged shapes and the fold path, several items per workgroup
emu.set_refine64(1)
try:
    for (ny, nx), ups in (((64, 64), (10,) if quick else (10, 20, 28, 43)), ((47, 61), () if quick else (10, 43)),
                          ((80, 80), () if quick else (10, 28)), ((85, 70), () if quick else (20, 59))):
        if not ups:
            continue
        tx, ty, sg, am = datagen.random_params(13, 3, max(ny, nx))
        ref = np.stack([datagen.pair_set(ny, nx, tx[k], ty[k], sg[k], am[k])[0] for k in range(3)])
        img = np.stack([datagen.pair_set(ny, nx, tx[k], ty[k], sg[k], am[k])[1] for k in range(3)])
        for up in ups:
            emu.set_grid(2)
            try:
                got, st = emu.pair(ref, img, up)
            finally:
                emu.set_grid(0)
            exp, est = orc.xcorr_refine_batch(ref, img, up)
            print('float64-refine pair %3dx%-3d U=%-2d max |d| %.2e' % (ny, nx, up, float(np.abs(got - exp).max())), flush=True)
            assert np.array_equal(st, est) and np.abs(got - exp).max() < 1e-4
finally:
    emu.set_refine64(-1)
print('sanitizer cases OK')
