#!/usr/bin/env python
"""Pair-mode throughput for cutout shapes that are not the full tile (staging takes the
general path there): pairs/s at upsample 10."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import datagen                                         # noqa: E402
import subpixal_amd as spx                             # noqa: E402

N = int(os.environ.get('N', 100000))
for ny, nx in ((64, 64), (64, 60), (63, 63), (48, 48), (40, 56), (33, 33), (32, 32), (24, 24), (17, 31), (65, 65), (72, 72), (85, 85),
               (80, 40), (86, 86), (96, 96), (100, 100), (128, 128)):
    ref, img, truth = datagen.pair_batch(3, 64, max(ny, nx))
    ref = np.ascontiguousarray(ref[:, :ny, :nx]); img = np.ascontiguousarray(img[:, :ny, :nx])
    n = N if max(ny, nx) <= 85 else N // 5
    r = torch.from_numpy(ref).cuda().repeat(n // 64, 1, 1).contiguous()
    m = torch.from_numpy(img).cuda().repeat(n // 64, 1, 1).contiguous()
    spx.xcorr_refine_batch(r, m, upsample=10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        d = spx.xcorr_refine_batch(r, m, upsample=10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print('%3dx%-3d %10.3e pairs/s  %7.1f GB/s' % (ny, nx, r.shape[0] / dt, r.shape[0] * ny * nx * 8 / dt / 1e9))
