"""ctypes access to the CPU logic-check build of the kernels (tests/cpu_emu).
TEST INFRASTRUCTURE: runs the same kernel source hipcc compiles, on CPU threads."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# SPX_EMU_LIB: another build of the same harness (tools/run_emu_asan.sh points it at the
# address-sanitizer one)
LIB = os.environ.get('SPX_EMU_LIB') or os.path.join(ROOT, 'tests', 'cpu_emu', 'libspx_emu.so')
_fp = ctypes.POINTER(ctypes.c_float)
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_bp = ctypes.POINTER(ctypes.c_uint8)
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ROOT, 'tests', 'cpu_emu', f) for f in ('emu_kernels.cpp', 'spx_rt_emu.h')]
        srcs += [os.path.join(ROOT, 'subpixal_amd', 'csrc', f)
                 for f in sorted(os.listdir(os.path.join(ROOT, 'subpixal_amd', 'csrc'))) if f.endswith('.h')]
        if not os.environ.get('SPX_EMU_LIB') and (
                not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)):
            subprocess.check_call(['make', '-C', os.path.join(ROOT, 'subpixal_amd', 'csrc'), 'emu'])
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def set_grid(g):
    lib().emu_set_grid(ctypes.c_int64(g))


def set_refine64(v):
    """refine arithmetic of the 64 tile and its fold path in pair(): -1 the product's default (float32:
    SPX_REFINE_DEFAULT), 0 float32 (SPX_REFINE_F32), 1 float64 (SPX_REFINE_F64)"""
    lib().emu_set_refine64(int(v))


def set_disp5_packed(v):
    """64-tile reference-mode kernel: 0 round 2's, 1 the product's rule (the five-transform kernel of
    spx_kernels5.h up to 64 px, the eight-transform one on the fold path), 2 always the five-transform one"""
    lib().emu_set_disp5_packed(int(v))


def first_item(b, nwg):
    fn = lib().emu_first_item
    fn.restype = ctypes.c_int64
    return int(fn(ctypes.c_int64(b), ctypes.c_int64(nwg)))


def _in_dtype(*arrays):
    """float64 arrays go to the float64-input kernels, everything else is float32"""
    return np.float64 if all(np.asarray(a).dtype == np.float64 for a in arrays) else np.float32


def pair(ref, img, upsample=1, cc=0, tile=0):
    """tile: 0 = the product's dispatch (spx_capi.hip), else force the 32 / 64 tile or the
    period-192 / 256 path"""
    dt = _in_dtype(ref, img)
    ref = np.ascontiguousarray(ref, dt)
    img = np.ascontiguousarray(img, dt)
    n = ref.shape[0]
    out = np.zeros((n, 2))
    st = np.zeros(n, np.int32)
    fn, pt = (lib().emu_pair_f64, _dp) if dt == np.float64 else (lib().emu_pair_f32, _fp)
    rc = fn(_p(ref, pt), _p(img, pt), ctypes.c_int64(n), ref.shape[1], ref.shape[2],
            int(upsample), int(cc), _p(out, _dp), _p(st, _ip), int(tile))
    assert rc == 0, rc
    return out, st


def disp5(ref, im4, cc=1):
    dt = _in_dtype(ref, im4)
    ref = np.ascontiguousarray(ref, dt)
    im4 = np.ascontiguousarray(im4, dt)
    n, ny, nx = ref.shape
    out = np.zeros((n, 2))
    st = np.zeros(n, np.int32)
    icc = np.zeros((n, 2 * ny, 2 * nx), np.float32)
    fn, pt = (lib().emu_disp5_f64, _dp) if dt == np.float64 else (lib().emu_disp5_f32, _fp)
    rc = fn(_p(ref, pt), _p(im4, pt), ctypes.c_int64(n), ny, nx, int(cc),
            _p(icc, _fp), _p(out, _dp), _p(st, _ip))
    assert rc == 0, rc
    return out, st, icc


def disp5_var(refs, im4s, family, cc=1):
    """variable-shape batch (float32): lists of 2-D refs and [4, ny, nx] dither stacks"""
    shapes = np.array([r.shape for r in refs], np.int32)
    sizes = shapes[:, 0].astype(np.int64) * shapes[:, 1]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    total = int(sizes.sum())
    ref = np.concatenate([np.asarray(r, np.float32).ravel() for r in refs])
    im4 = np.concatenate([np.asarray(m, np.float32).ravel() for m in im4s])
    n = len(refs)
    out = np.zeros((n, 2))
    st = np.zeros(n, np.int32)
    icc = np.zeros(4 * total, np.float32)
    i64 = ctypes.POINTER(ctypes.c_int64)
    rc = lib().emu_disp5_var_f32(_p(ref, _fp), _p(im4, _fp), offs.ctypes.data_as(i64),
                                 shapes.ctypes.data_as(_ip), ctypes.c_int64(n), int(family), int(cc),
                                 _p(icc, _fp), _p(out, _dp), _p(st, _ip))
    assert rc == 0, rc
    iccs = [icc[4 * int(o):4 * int(o) + 4 * int(s)].reshape(2 * sh[0], 2 * sh[1])
            for o, s, sh in zip(offs, sizes, shapes)]
    return out, st, iccs


def find_peak(images, guesses=None, fit=(5, 5), search=(0, 0), masks=None):
    images = np.ascontiguousarray(images, np.float64)
    n, ny, nx = images.shape
    if masks is not None:
        masks = np.ascontiguousarray(masks, np.uint8)
    if guesses is not None:
        guesses = np.ascontiguousarray(guesses, np.float64)
    out = np.zeros((n, 2))
    st = np.zeros(n, np.int32)
    rc = lib().emu_find_peak(_p(images, _dp), _p(masks, _bp), _p(guesses, _dp), ctypes.c_int64(n),
                             ny, nx, fit[0], fit[1], search[0], search[1], _p(out, _dp), _p(st, _ip))
    assert rc == 0, rc
    return out, st


def blot_affine4(src, affine, ny, nx, gain=None):
    src = np.ascontiguousarray(src, np.float32)
    affine = np.ascontiguousarray(affine, np.float64)
    n = src.shape[0]
    im4 = np.zeros((n, 4, ny, nx), np.float32)
    if gain is not None:
        gain = np.ascontiguousarray(gain, np.float32)
    rc = lib().emu_blot_affine4(_p(src, _fp), ctypes.c_int64(n), src.shape[1], src.shape[2],
                                affine.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _p(gain, _fp),
                                ny, nx, _p(im4, _fp))
    assert rc == 0, rc
    return im4


def blot_poly4(src, coef, degree, ny, nx, gain=None):
    src = np.ascontiguousarray(src, np.float32)
    coef = np.ascontiguousarray(coef, np.float64)
    n = src.shape[0]
    im4 = np.zeros((n, 4, ny, nx), np.float32)
    if gain is not None:
        gain = np.ascontiguousarray(gain, np.float32)
    rc = lib().emu_blot_poly4(_p(src, _fp), ctypes.c_int64(n), src.shape[1], src.shape[2],
                              coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(degree),
                              _p(gain, _fp), ny, nx, _p(im4, _fp))
    assert rc == 0, rc
    return im4


def label_bboxes(seg, max_label):
    seg = np.ascontiguousarray(seg, np.int32)
    boxes = np.zeros((max_label + 1, 4), np.int32)
    counts = np.zeros(max_label + 1, np.int32)
    i32 = ctypes.POINTER(ctypes.c_int32)
    rc = lib().emu_label_bboxes(seg.ctypes.data_as(i32), seg.shape[0], seg.shape[1], int(max_label),
                                boxes.ctypes.data_as(i32), counts.ctypes.data_as(i32))
    assert rc == 0
    return boxes, counts


def gather(frame, fmask, boxes, tny, tnx, fill, seg=None, ids=None):
    frame = np.ascontiguousarray(frame, np.float32)
    boxes = np.ascontiguousarray(boxes, np.int32)
    if fmask is not None:
        fmask = np.ascontiguousarray(fmask, np.uint8)
    n = boxes.shape[0]
    tiles = np.zeros((n, tny, tnx), np.float32)
    rc = lib().emu_gather(_p(frame, _fp), _p(fmask, _bp), frame.shape[0], frame.shape[1],
                          boxes.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_int64(n),
                          tny, tnx, ctypes.c_float(fill), _p(tiles, _fp),
                          None if seg is None else np.ascontiguousarray(seg, np.int32).ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                          None if ids is None else np.ascontiguousarray(ids, np.int32).ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert rc == 0
    return tiles


def gen_pairs(seed, first, count, n, slo, shi, maxshift):
    ref = np.zeros((count, n, n), np.float32)
    img = np.zeros_like(ref)
    truth = np.zeros((count, 2))
    rc = lib().emu_gen_pairs(ctypes.c_uint64(seed), ctypes.c_int64(first), ctypes.c_int64(count), n,
                             ctypes.c_float(slo), ctypes.c_float(shi), ctypes.c_float(maxshift),
                             _p(ref, _fp), _p(img, _fp), _p(truth, _dp))
    assert rc == 0
    return ref, img, truth


def gather_var(frame, fmask, boxes, fill, seg=None, ids=None):
    """variable-shape gather: returns (packed float32, offsets int64)"""
    frame = np.ascontiguousarray(frame, np.float32)
    boxes = np.ascontiguousarray(boxes, np.int32)
    sizes = boxes[:, 2].astype(np.int64) * boxes[:, 3]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    out = np.full(int(sizes.sum()), -7.0, np.float32)
    i32, i64 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    if fmask is not None:
        fmask = np.ascontiguousarray(fmask, np.uint8)
    seg_a = None if seg is None else np.ascontiguousarray(seg, np.int32)
    ids_a = None if ids is None else np.ascontiguousarray(ids, np.int32)
    rc = lib().emu_gather_var(_p(frame, _fp), _p(fmask, _bp), frame.shape[0], frame.shape[1], boxes.ctypes.data_as(i32),
                              ctypes.c_int64(len(boxes)), offs.ctypes.data_as(i64), ctypes.c_float(fill), _p(out, _fp),
                              None if seg_a is None else seg_a.ctypes.data_as(i32),
                              None if ids_a is None else ids_a.ctypes.data_as(i32))
    assert rc == 0
    return out, offs


def blot4_var(src, src_offs, src_shapes, maps, degree, dst_shapes, gain=None):
    """variable-shape blots: returns (im4 packed float32, dst offsets)"""
    src = np.ascontiguousarray(src, np.float32)
    src_offs = np.ascontiguousarray(src_offs, np.int64)
    src_shapes = np.ascontiguousarray(src_shapes, np.int32)
    dst_shapes = np.ascontiguousarray(dst_shapes, np.int32)
    maps = np.ascontiguousarray(maps, np.float64)
    sizes = dst_shapes[:, 0].astype(np.int64) * dst_shapes[:, 1]
    doffs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    im4 = np.full(4 * int(sizes.sum()), -7.0, np.float32)
    if gain is not None:
        gain = np.ascontiguousarray(gain, np.float32)
    i32, i64 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    rc = lib().emu_blot4_var(_p(src, _fp), src_offs.ctypes.data_as(i64), src_shapes.ctypes.data_as(i32),
                             ctypes.c_int64(len(src_offs)), maps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                             int(degree), _p(gain, _fp), doffs.ctypes.data_as(i64), dst_shapes.ctypes.data_as(i32),
                             _p(im4, _fp))
    assert rc == 0
    return im4, doffs
