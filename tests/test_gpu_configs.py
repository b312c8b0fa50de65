"""BASELINE.json configs 4 and 5 at their stated per-GPU size.

configs[3]: 1e7 64x64 pairs sharded over 8 GPUs -> rank 0's shard, dist.shard_range(10**7, 0, 8)
= 1.25e6 pairs (41 GB of inputs, generated on the device), upsample=10.
configs[4]: end-to-end alignment of a synthetic 4096x4096 frame pair with a 5000-source catalog,
64x64 cutouts (one GPU takes the whole catalog here; with 8 ranks each takes 625 sources)."""
import os
import sys

import numpy as np
import pytest

from oracle import subpixal_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_config4_rank0_shard_of_1e7_pairs():
    import torch
    import subpixal_amd
    from subpixal_amd import synth
    from subpixal_amd.dist import shard_range
    lo, hi = shard_range(10 ** 7, 0, 8)
    assert (lo, hi) == (0, 1250000)
    assert shard_range(10 ** 7, 7, 8) == (8750000, 10 ** 7)
    n = hi - lo
    # generated in four slices (the generator's scratch scales with the slice), same stream of
    # pairs as one call: pair k depends on (seed, k) only
    ref = torch.empty((n, 64, 64), dtype=torch.float32, device='cuda')
    img = torch.empty_like(ref)
    truth = torch.empty((n, 2), dtype=torch.float64, device='cuda')
    step = n // 4
    for s in range(0, n, step):
        e = min(n, s + step)
        r, i, t = synth.gaussian_pairs(e - s, 64, seed=20261003, first_index=lo + s)
        ref[s:e], img[s:e], truth[s:e] = r, i, t
        del r, i, t
    torch.cuda.synchronize()
    d1, st = subpixal_amd.xcorr_refine_batch(ref, img, upsample=10, return_status=True)
    torch.cuda.synchronize()
    assert d1.shape == (n, 2)
    assert float((d1 - truth).abs().max()) < 1e-3            # every one of the 1.25e6 pairs
    assert int(st.abs().max()) == 0
    d2 = subpixal_amd.xcorr_refine_batch(ref, img, upsample=10)
    assert torch.equal(d1, d2)                               # bitwise determinism at full size
    # the first 1e5 pairs of this shard are bench.py's configs[1] batch: same numbers
    d3 = subpixal_amd.xcorr_refine_batch(ref[:100000], img[:100000], upsample=10)
    assert torch.equal(d3, d1[:100000])
    # oracle on a bounded sample spread over the shard
    pick = np.linspace(0, n - 1, 24).astype(np.int64)
    exp, est = orc.xcorr_refine_batch(ref[pick].cpu().numpy(), img[pick].cpu().numpy(), 10)
    assert np.max(np.abs(d1[pick].cpu().numpy() - exp)) < 2e-4
    assert np.array_equal(st[pick].cpu().numpy(), est)


def test_config5_full_size_alignment():
    import align_synthetic
    # crowded field (5000 sources on 4096^2: one cutout in ten holds part of a neighbour): let the
    # sigma clipping converge instead of stopping after the reference's default 3 rounds
    out = align_synthetic.run(size=4096, nsrc=5000, upsample=10, quiet=True, nclip=12)
    err = np.abs(out['shifts'] - out['true_shifts']).max(axis=1)
    fit = out['fit']
    print('config 5: median |d| %.3g px, kept %d/5000, offset err %s, matrix err %.3g, gpu %.1f ms'
          % (np.median(err), fit['fitmask'].sum(), fit['offset'] - out['true_offset'],
             np.abs(fit['fit_matrix'] - out['true_matrix']).max(), 1e3 * out['gpu_seconds']))
    # crowding: with 5000 sources on 4096^2 some cutouts hold a neighbour; the sigma-clipped fit
    # has to recover the transform all the same (bounds as test_gpu_align.py)
    assert np.median(err) < 5e-4
    assert fit['fitmask'].sum() > 4000
    assert np.abs(fit['offset'] - out['true_offset']).max() < 1e-3
    assert np.abs(fit['fit_matrix'] - out['true_matrix']).max() < 3e-6
