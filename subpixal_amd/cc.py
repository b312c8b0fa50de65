"""Sub-pixel cross-correlation displacements on the MI355X.

Drop-in for ``subpixal.cc`` (reference: /root/reference/subpixal/cc.py):
``find_displacement`` keeps the reference signature and return values
(cc.py:21-95); the batched forms run the whole per-source loop of
``align.find_linear_fit`` (align.py:656-699) in one kernel launch.  All
arithmetic happens in libsubpixal_hip.so; there is no CPU path here.
"""
import numpy as np
import torch

from . import _ffi, device

__all__ = ['find_displacement', 'find_displacement_batch', 'xcorr_refine_batch']


def _cc_code(cc_type):
    # cc.py:107-108: upper-cased; anything but NCC/ZNCC means plain CC
    return _ffi.CC_CODES.get(str(cc_type).upper(), 0)


def _workspace(nbytes, dev):
    """Device scratch for cutouts above 64 px (the 128 tile keeps its class planes in an
    L2-resident workspace); (None, 0) when the call needs none."""
    if nbytes == 0:
        return None, 0
    return torch.empty((int(nbytes),), dtype=torch.uint8, device=dev), int(nbytes)


def _finish(t, like_torch):
    return t if like_torch else t.cpu().numpy()


def _input_dtype(*arrays):
    """float64 inputs keep their type on the device (the ``_f64`` entry points build the
    ``!= 0`` masks and the statistics of cc.py:131-156 from the float64 values, as the
    reference does for float64 cutouts); everything else is computed from float32."""
    def is64(a):
        return (a.dtype == torch.float64) if isinstance(a, torch.Tensor) else \
            (np.asarray(a).dtype == np.float64)
    return torch.float64 if all(is64(a) for a in arrays) else torch.float32


def xcorr_refine_batch(ref, img, upsample=1, cc_type='CC', return_status=False):
    """Shifts of ``img[k]`` relative to ``ref[k]`` for a batch of cutout pairs.

    ref, img : ``[N, ny, nx]`` float32 or float64, torch CUDA tensors (used in place) or
        numpy arrays (copied to the device).  5 <= ny, nx <= 128.  float64 pairs are masked
        and normalised in float64 (cc.py:131-156) before the float32 transforms.
    upsample : the cross-correlation is refined on a grid ``upsample`` times
        finer than the pixel grid before the 5x5 quadratic peak fit;
        ``upsample=2`` is the reference's half-pixel interlace (cc.py:121-126),
        ``upsample=1`` is ``fftconvolve(..,'same')`` + ``find_peak`` on one image.

    Returns ``dxdy [N, 2]`` float64 (torch CUDA tensor if the inputs were
    tensors, else numpy) and, with ``return_status``, the int32 ``status [N]``.
    """
    like_torch = isinstance(ref, torch.Tensor)
    dt = _input_dtype(ref, img)
    r = device.to_device(ref, dt)
    m = device.to_device(img, dt)
    if r.dim() != 3 or r.shape != m.shape:
        raise ValueError("ref and img must both have shape [N, ny, nx].")
    n, ny, nx = r.shape
    out = torch.empty((n, 2), dtype=torch.float64, device=r.device)
    status = torch.empty((n,), dtype=torch.int32, device=r.device)
    lib = _ffi.load()
    with torch.cuda.device(r.device):
        ws, ws_bytes = _workspace(lib.spx_workspace_bytes_xcorr(n, ny, nx), r.device)
        fn = lib.spx_xcorr_refine_f64 if dt == torch.float64 else lib.spx_xcorr_refine_f32
        _ffi.check(fn(
            device.ptr(r), device.ptr(m), n, ny, nx, int(upsample), _cc_code(cc_type),
            device.ptr(out), device.ptr(status), device.ptr(ws), ws_bytes, device.stream_ptr()))
    if return_status:
        return _finish(out, like_torch), _finish(status, like_torch)
    return _finish(out, like_torch)


def find_displacement_batch(ref, im4, cc_type='NCC', full_output=False, return_status=False):
    """``find_displacement`` for a batch: ``ref [N, ny, nx]``, ``im4 [N, 4, ny, nx]``
    (image00, image10, image01, image11).  Returns ``dxdy [N, 2]`` float64, plus the
    interlaced images ``icc [N, 2ny, 2nx]`` with ``full_output`` and the status
    array with ``return_status``."""
    like_torch = isinstance(ref, torch.Tensor)
    dt = _input_dtype(ref, im4)
    r = device.to_device(ref, dt)
    m = device.to_device(im4, dt)
    if r.dim() != 3 or m.dim() != 4 or m.shape[1] != 4 or \
            (r.shape[0],) + tuple(r.shape[1:]) != (m.shape[0],) + tuple(m.shape[2:]):
        raise ValueError("All cutouts must have same shape.")
    n, ny, nx = r.shape
    out = torch.empty((n, 2), dtype=torch.float64, device=r.device)
    status = torch.empty((n,), dtype=torch.int32, device=r.device)
    icc = torch.empty((n, 2 * ny, 2 * nx), dtype=torch.float32, device=r.device)
    lib = _ffi.load()
    with torch.cuda.device(r.device):
        ws, ws_bytes = _workspace(lib.spx_workspace_bytes_displacement5(n, ny, nx, 0), r.device)
        fn = lib.spx_find_displacement5_f64 if dt == torch.float64 else lib.spx_find_displacement5_f32
        _ffi.check(fn(
            device.ptr(r), device.ptr(m), n, ny, nx, _cc_code(cc_type), device.ptr(out),
            device.ptr(status), device.ptr(icc), device.ptr(ws), ws_bytes, device.stream_ptr()))
    res = [_finish(out, like_torch)]
    if full_output:
        res.append(_finish(icc, like_torch))
    if return_status:
        res.append(_finish(status, like_torch))
    return res[0] if len(res) == 1 else tuple(res)


def find_displacement(ref_image, image00, image10, image01, image11,
                      cc_type='NCC', full_output=False):
    """Find subpixel displacements between one reference cutout and a set of four
    "dithered" cutouts from the peak of the interlaced cross-correlation image.

    Same signature, return values and error behaviour as the reference
    ``subpixal.cc.find_displacement`` (cc.py:21-95); the transforms run in float32 on the GPU,
    float64 cutouts are masked and normalised in float64 first (cc.py:131-156).

    Returns ``(dx, dy)`` or, with ``full_output``, ``(dx, dy, icc, ccs)``.
    """
    ims = [np.asarray(a) for a in (ref_image, image00, image10, image01, image11)]
    if not all(im.shape == ims[0].shape for im in ims) or ims[0].ndim != 2:
        raise ValueError("All cutouts must have same shape.")      # cc.py:103-105
    dt = np.float64 if all(im.dtype == np.float64 for im in ims) else np.float32
    ref = np.ascontiguousarray(ims[0], dtype=dt)[None]
    im4 = np.ascontiguousarray(np.stack(ims[1:]), dtype=dt)[None]
    dxdy, icc = find_displacement_batch(ref, im4, cc_type=cc_type, full_output=True)
    dx, dy = np.float64(dxdy[0, 0]), np.float64(dxdy[0, 1])
    if not full_output:
        return dx, dy
    icc = icc[0]
    ccs = (icc[0::2, 0::2][::-1, ::-1], icc[0::2, 1::2][::-1, ::-1],
           icc[1::2, 0::2][::-1, ::-1], icc[1::2, 1::2][::-1, ::-1])   # cc.py:121-126
    return dx, dy, icc, ccs
