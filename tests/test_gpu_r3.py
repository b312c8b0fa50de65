"""Round-3 parity tests (`-m gpu`): the two carve-outs of the round-2 sweeps as tests, and the
eight-wave A/B kernel of the 64 tile through the C-ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

import datagen
from oracle import subpixal_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def spx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import subpixal_amd
    return subpixal_amd


def _catalog(rng, count, wide):
    """sources of one shape each; `wide`: 100..140 px cutouts with spots of n/10..n/6 px -- the kind on
    which the reference's own 5x5 fit is ill-conditioned (centroid.py:219-234 accepts the quadratic's
    stationary point anywhere inside the image; tools/sweep_disp5.py set 10 of 115,200 such sources apart)"""
    refs, ims = [], []
    for _ in range(count):
        if wide:
            ny, nx = int(rng.integers(96, 141)), int(rng.integers(93, 141))
            sigma = min(ny, nx) / rng.uniform(6, 10)
        else:
            ny, nx = int(rng.integers(3, 129)), int(rng.integers(3, 129))
            sigma = max(0.8, min(ny, nx) / rng.uniform(8, 14))
        smax = min(2.0, min(ny, nx) / 6.0)
        dt = np.float64 if rng.random() < 0.3 else np.float32
        t = datagen.dither_set(ny, nx, rng.uniform(-smax, smax), rng.uniform(-smax, smax), sigma,
                               rng.uniform(0.5, 2.0), dt, int(rng.integers(1, 1 << 30)) if rng.random() < 0.7 else 0,
                               float(rng.choice([0.003, 0.02])), int(rng.choice([0, 0, 1, 2])) if min(ny, nx) >= 12 else 0)
        refs.append(t[0])
        ims.append(np.stack(t[1:]))
    return refs, ims


def test_fit_stage_is_exact_on_the_gpus_own_interlaced_image(spx):
    """centroid.find_peak (centroid.py:18-236 with cc.py:86's arguments) applied by the ORACLE to the
    interlaced image the GPU returned (full_output) must give the GPU's (dx, dy, status): that proves the
    arg-max + fit + every early return exact, so that any difference to the float64 reference path is the
    3e-7 relative noise of the float32 transforms in the image -- including the sources whose fit is
    ill-conditioned in the reference itself (stationary point outside the 5x5 box it was fitted on)."""
    rng = np.random.default_rng(20261004)
    outside = 0
    for name, wide in (('CC', False), ('NCC', False), ('ZNCC', False), ('CC', True), ('NCC', True), ('ZNCC', True)):
        refs, ims = _catalog(rng, 64, wide)
        d, iccs, st = spx.find_displacement_var(refs, ims, cc_type=name, full_output=True, return_status=True)
        for k in range(len(refs)):
            icc = np.asarray(iccs[k], np.float64)           # the GPU's float32 values, exactly
            s2 = []
            xm, ym = orc.find_peak(icc, peak_fit_box=5, peak_search_box='all', _status=s2)
            ex = 0.5 * xm - (icc.shape[1] - 1) // 4         # cc.py:89-93
            ey = 0.5 * ym - (icc.shape[0] - 1) // 4
            assert st[k] == s2[-1], (k, name, refs[k].shape, st[k], s2[-1])
            # the same box through the constant operator the kernel implements
            x5, y5, s5 = orc.find_peak_5x5_all(icc)
            assert s5 == st[k]
            jm, im_ = np.unravel_index(int(np.argmax(icc)), icc.shape)
            far = max(abs(xm - im_), abs(ym - jm)) > 2.5   # vertex outside the box it was fitted on
            outside += bool(far)
            # float64 against float64: 1e-9 px; a vertex far outside its box is a ratio of two nearly
            # cancelling float64 quantities, where two correct float64 evaluations differ by more
            tol = 1e-6 if far else 1e-9
            assert abs(d[k, 0] - (0.5 * x5 - (icc.shape[1] - 1) // 4)) <= tol, (k, name, refs[k].shape)
            assert abs(d[k, 1] - (0.5 * y5 - (icc.shape[0] - 1) // 4)) <= tol, (k, name, refs[k].shape)
            assert abs(d[k, 0] - ex) <= 100 * tol and abs(d[k, 1] - ey) <= 100 * tol, (k, name, refs[k].shape)
    print('sources with the fitted maximum outside its 5x5 box: %d of 384' % outside)


_EIGHT_WAVE_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests')
import datagen, subpixal_amd as spx
from subpixal_amd import synth
from oracle import subpixal_oracle as orc
for up, tol in ((1, 1e-5), (2, 2e-5), (10, 2e-4), (20, 2e-4)):
    ref, img, truth = datagen.pair_batch(21, 24, 64)
    got, st = spx.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine='float32')   # (the eight-wave kernel has no float64 refine: the default would hand upsample >= 28 to the four-wave kernel)
    exp, est = orc.xcorr_refine_batch(ref, img, up)
    assert np.max(np.abs(got - exp)) < tol, (up, np.max(np.abs(got - exp)))
    assert np.array_equal(st, est)
rng = np.random.default_rng(5)
for ny, nx, up, name in ((63, 61, 10, 'NCC'), (40, 57, 2, 'ZNCC'), (5, 5, 1, 'CC'), (64, 37, 30, 'NCC'), (33, 64, 45, 'CC')):
    r = np.empty((6, ny, nx), np.float32); i = np.empty_like(r)
    for k in range(6):
        tx, ty = rng.uniform(-2, 2, 2) if min(ny, nx) > 8 else rng.uniform(-0.5, 0.5, 2)
        r[k], i[k] = datagen.pair_set(ny, nx, tx, ty, rng.uniform(1.2, min(ny, nx) / 8 + 1.2), 1.3, np.float32,
                                      noise_seed=int(rng.integers(1, 1000)), noise_level=0.01)
    got, st = spx.xcorr_refine_batch(r, i, upsample=up, cc_type=name, return_status=True, refine='float32')
    exp, est = orc.xcorr_refine_batch(r, i, up, name)
    assert np.max(np.abs(got - exp)) < 3e-4 and np.array_equal(st, est), (ny, nx, up, name)
ref, img, truth = synth.gaussian_pairs(30000, 64, seed=99)
d1, st = spx.xcorr_refine_batch(ref, img, upsample=10, return_status=True)
assert float((d1 - truth).abs().max()) < 1e-3 and int(st.abs().max()) == 0
assert torch.equal(d1, spx.xcorr_refine_batch(ref, img, upsample=10))
perm = torch.randperm(30000, device=ref.device)[:5000]
assert torch.equal(spx.xcorr_refine_batch(ref[perm].contiguous(), img[perm].contiguous(), upsample=10), d1[perm])
print('eight-wave kernel OK')
'''


def test_eight_wave_kernel_through_the_c_abi(spx):
    """SPX_PAIR64_WAVES=8 (read once per process, hence the child): the 16-class / half-wave kernel of
    spx_kernels8.h against the oracle on full and ragged tiles, every cc_type and window size, plus
    truth / determinism / permutation properties on 3e4 device-generated pairs."""
    env = dict(os.environ, SPX_PAIR64_WAVES='8')
    out = subprocess.run([sys.executable, '-c', _EIGHT_WAVE_CHILD % {'root': ROOT}], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'eight-wave kernel OK' in out.stdout


_PACKED_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests')
import datagen, subpixal_amd as spx
from oracle import subpixal_oracle as orc
g = np.load(%(root)r + '/tests/golden/find_displacement.npz', allow_pickle=False)
for n, shape, name, dt in ((64, None, 'NCC', np.float32), (64, None, 'CC', np.float32), (48, None, 'ZNCC', np.float64),
                           (77, None, 'NCC', np.float32), (85, None, 'CC', np.float64), (40, (33, 40), 'ZNCC', np.float32),
                           (5, (3, 5), 'CC', np.float32)):
    r5, m4, _ = datagen.dither_batch(7, 6, n, dtype=dt)
    if shape:
        r5 = np.ascontiguousarray(r5[:, :shape[0], :shape[1]]); m4 = np.ascontiguousarray(m4[:, :, :shape[0], :shape[1]])
    d, icc, st = spx.find_displacement_batch(r5, m4, cc_type=name, full_output=True, return_status=True)
    e, est = orc.find_displacement_batch(r5, m4, name)
    assert np.array_equal(st, est), (n, name)
    assert np.abs(d - e).max() < 3e-5, (n, name, np.abs(d - e).max())
    eicc = orc.build_icc(r5[0], *m4[0], cc_type=name)[0]
    assert np.abs(icc[0] - eicc).max() < 3e-6 * np.abs(eicc).max()
print('five-transform kernel OK')
'''


def test_five_transform_reference_mode_kernel_for_every_cc_type(spx):
    """SPX_DISP5_PACKED=2 (read once per process, hence the child): spx_kernels5.h takes every 64-tile
    reference-mode call -- the fold path (65..85 px) too, which the default dispatch leaves to the eight-transform
    kernel -- for every cc_type, float64 inputs and ragged shapes, against the oracle's cc.find_displacement
    (cc.py:21-95)."""
    env = dict(os.environ, SPX_DISP5_PACKED='2')
    out = subprocess.run([sys.executable, '-c', _PACKED_CHILD % {'root': ROOT}], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'five-transform kernel OK' in out.stdout


@pytest.mark.gpu
def test_float64_refine_through_the_c_abi():
    """spx_xcorr_refine_ex_* with SPX_REFINE_F64 (cc.xcorr_refine_batch(refine='float64')): the 64 tile and its
    fold path accumulate the refine in float64 -- measured 1.2e-5 / 2.3e-5 px (64 px) and 1.4e-5 / 2.7e-5 px
    (80 px) against the float64 definition at upsample 10 / 20 where the default is at 5.5e-5 ... 1.3e-4
    (profiles/r03/refine_precision*.txt).  Families without a second form return the same bits either way."""
    from subpixal_amd import cc
    for n, bound in ((64, 6e-5), (80, 6e-5)):
        ref, img, _ = datagen.pair_batch(20261005 + n, 24, n)
        for up in (10, 20):
            exp, est = orc.xcorr_refine_batch(ref, img, up)
            got64, st64 = cc.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine='float64')
            got32, st32 = cc.xcorr_refine_batch(ref, img, upsample=up, return_status=True)
            assert np.array_equal(st64, est) and np.array_equal(st32, est)
            d64, d32 = np.abs(got64 - exp).max(), np.abs(got32 - exp).max()
            assert d64 < bound, (n, up, d64)
            assert d32 < 2e-4, (n, up, d32)                  # the default's bound, unchanged
            assert d64 < d32, (n, up, d64, d32)
    for n in (32, 96):                                       # float32 only / float64 only: one form each
        ref, img, _ = datagen.pair_batch(7, 8, n)
        a = cc.xcorr_refine_batch(ref, img, upsample=10, refine='float64')
        b = cc.xcorr_refine_batch(ref, img, upsample=10)
        c = cc.xcorr_refine_batch(ref, img, upsample=10, refine='float32')
        assert np.array_equal(a, b) and np.array_equal(a, c), n
    with pytest.raises(ValueError):
        cc.xcorr_refine_batch(ref, img, upsample=10, refine='float128')
    # float64 cutouts (spx_xcorr_refine_ex_f64) take the same two forms
    for n in (64, 80):
        ref, img, _ = datagen.pair_batch(20261005 + n, 12, n, dtype=np.float64)
        for up in (10, 28):
            exp, est = orc.xcorr_refine_batch(ref, img, up)
            got64, st64 = cc.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine='float64')
            got32 = cc.xcorr_refine_batch(ref, img, upsample=up)
            assert np.array_equal(st64, est)
            assert np.abs(got64 - exp).max() < 6e-5 and np.abs(got32 - exp).max() < 3e-4, (n, up)
            assert np.abs(got64 - exp).max() < np.abs(got32 - exp).max(), (n, up)
    # the default on 33..85 px is the float32 form at every upsample (spx_capi.hip refine64_is_f64: why)
    for n in (64, 80):
        ref, img, _ = datagen.pair_batch(3, 8, n)
        for up in (10, 27, 28, 43, 59):
            d = cc.xcorr_refine_batch(ref, img, upsample=up)
            assert np.array_equal(d, cc.xcorr_refine_batch(ref, img, upsample=up, refine='float32')), (n, up)
            assert not np.array_equal(d, cc.xcorr_refine_batch(ref, img, upsample=up, refine='float64')), (n, up)


@pytest.mark.gpu
def test_wide_spots_and_the_two_refine_forms():
    """The accuracy domain of each refine form on the 64 tile and its fold path, as include/subpixal_hip.h states it
    (profiles/r03/width_precision_256.txt, 256 pairs per cell): spots of sigma 8..11 px are within 1e-3 px of the
    float64 definition in the DEFAULT (float32) form up to upsample 39; spots of sigma 11..15 px are in the float32
    form up to upsample 27 only (2 % beyond at 39, 10..12 % at 59, worst 4.0e-3 px) and in the float64 form at every
    upsample (worst 5.3e-4 px).  (Sources that fill their cutout, sigma > side / 6, lose pairs in both: not asserted.)"""
    from subpixal_amd import cc
    def spots(n, lo, hi, count=64):
        tx, ty, sg, am = datagen.random_params(41, count, n, sigma_lo=lo, sigma_hi=hi)
        prs = [datagen.pair_set(n, n, tx[k], ty[k], sg[k], am[k]) for k in range(count)]
        return np.stack([p[0] for p in prs]), np.stack([p[1] for p in prs])
    for n in (64, 85):
        ref, img = spots(n, 8.0, 11.0)
        for up in (20, 39):
            exp, est = orc.xcorr_refine_batch(ref, img, up)
            got, st = cc.xcorr_refine_batch(ref, img, upsample=up, return_status=True)
            assert np.array_equal(st, est) and np.abs(got - exp).max() < 1e-3, (n, up, np.abs(got - exp).max())
        ref, img = spots(n, 11.0, 15.0)
        for up in (20, 27):
            exp, est = orc.xcorr_refine_batch(ref, img, up)
            got = cc.xcorr_refine_batch(ref, img, upsample=up)
            assert np.abs(got - exp).max() < 1e-3, (n, up, np.abs(got - exp).max())
        for up in (39, 59):
            exp, est = orc.xcorr_refine_batch(ref, img, up)
            got, st = cc.xcorr_refine_batch(ref, img, upsample=up, return_status=True, refine='float64')
            assert np.array_equal(st, est) and np.abs(got - exp).max() < 1e-3, (n, up, np.abs(got - exp).max())
        f32 = cc.xcorr_refine_batch(ref, img, upsample=59)
        print('%d px, sigma 11..15 px, upsample 59: float32 (default) %d of 64 pairs beyond 1e-3 px (worst %.2e), float64 worst %.2e'
              % (n, int((np.abs(f32 - exp).max(axis=1) > 1e-3).sum()), np.abs(f32 - exp).max(), np.abs(got - exp).max()))
