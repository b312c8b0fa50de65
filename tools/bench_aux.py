#!/usr/bin/env python
"""Measured rates of the kernels either side of the hot path, each against
its HBM roofline (algorithmic bytes / time; 8 TB/s peak).  BASELINE config-5 shapes: a 4096^2
frame, 5000 sources, 64x64 cutouts."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from subpixal_amd import blot, centroid, cutout, device       # noqa: E402

device.init()
PEAK = 8000.0


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e-3


def report(name, units, unit_name, nbytes, dt):
    gbs = nbytes / dt / 1e9
    print('%-28s %9.3f ms  %10.3e %s/s  %8.1f GB/s algorithmic = %5.2f %% of HBM peak'
          % (name, dt * 1e3, units / dt, unit_name, gbs, 100 * gbs / PEAK))


rng = np.random.default_rng(0)
S, NSRC, T = 4096, 5000, 64
frame = torch.as_tensor(rng.normal(size=(S, S)).astype(np.float32)).cuda()
mask = torch.zeros((S, S), dtype=torch.uint8, device='cuda')
seg_np = np.zeros((S, S), np.int32)
xy = rng.uniform(40, S - 40, (NSRC, 2)).astype(int)
for k, (x, y) in enumerate(xy):
    seg_np[y - 6:y + 7, x - 6:x + 7] = k + 1
seg = torch.as_tensor(seg_np).cuda()
boxes = torch.as_tensor(np.stack([xy[:, 0] - T // 2, xy[:, 1] - T // 2, np.full(NSRC, T), np.full(NSRC, T)],
                                 axis=1).astype(np.int32)).cuda()
ids = torch.arange(1, NSRC + 1, dtype=torch.int32, device='cuda')

# 8f-3: bounding boxes of all segments, one pass over the label image (4 B/pixel read)
dt = timed(lambda: cutout.segment_bounding_boxes(seg, NSRC))
report('label bboxes 4096^2', S * S, 'pixels', 4 * S * S, dt)

# 8f-1: cutout packing (read window + mask + labels, write tile)
dt = timed(lambda: cutout.pack_cutouts(frame, boxes, (T, T), mask=mask, segmentation_image=seg, ids=ids))
report('gather 5000 x 64x64', NSRC, 'cutouts', NSRC * T * T * (4 + 1 + 4 + 4), dt)
dt = timed(lambda: cutout.pack_cutouts(frame, boxes, (T, T)))
report('gather (no mask/labels)', NSRC, 'cutouts', NSRC * T * T * 8, dt)

# 8f-2: four dithered blots per source from an 80x80 drizzled cutout
src = torch.as_tensor(rng.normal(size=(NSRC, 80, 80)).astype(np.float32)).cuda()
aff = torch.as_tensor(blot.shift_affine(NSRC, 8.2, 7.6)).cuda()
dt = timed(lambda: blot.blot_affine4_batch(src, aff, (T, T)))
report('blot x4 5000 x 64x64', NSRC, 'sources', NSRC * (80 * 80 * 4 + 4 * T * T * 4), dt)

# general find_peak (float64 images, whole-image search, 5x5 fit)
imgs = torch.as_tensor(rng.normal(size=(NSRC, 2 * T, 2 * T))).cuda()
dt = timed(lambda: centroid.find_peak_batch(imgs, peak_fit_box=5, peak_search_box='all'), reps=5)
report('find_peak 5000 x 128x128 f64', NSRC, 'peaks', NSRC * 4 * T * T * 8, dt)
