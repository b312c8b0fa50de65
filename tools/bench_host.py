#!/usr/bin/env python3
"""PCIe-inclusive rate of the pair path when the caller hands over HOST arrays (numpy in,
numpy out through subpixal_amd.cc.xcorr_refine_batch): never bench.py's `value`, recorded in
DESIGN.md section 4.  Pageable and pinned host memory."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subpixal_amd import cc, device, synth      # noqa: E402

N, n, U = 50000, 64, 10
device.init()
ref, img, truth = synth.gaussian_pairs(N, n)
href, himg = ref.cpu().numpy(), img.cpu().numpy()
cc.xcorr_refine_batch(href[:100], himg[:100], upsample=U)      # tables, first-touch
for label, make in (('pageable', lambda a: a),
                    ('pinned', lambda a: torch.as_tensor(a).pin_memory())):
    a, b = make(href), make(himg)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if label == 'pinned':
            out = cc.xcorr_refine_batch(a.cuda(non_blocking=True), b.cuda(non_blocking=True), upsample=U).cpu()
        else:
            out = cc.xcorr_refine_batch(a, b, upsample=U)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    err = float(np.abs(np.asarray(out) - truth.cpu().numpy()).max())
    gb = 2 * N * n * n * 4 / 1e9
    print('%s host arrays: %d pairs in %.1f ms = %.2e pairs/s (%.1f GB/s over PCIe), max err %.1e px'
          % (label, N, best * 1e3, N / best, gb / best, err))
