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

__all__ = ['find_displacement', 'find_displacement_batch', 'find_displacement_var', 'find_displacement_packed',
           'xcorr_refine_batch']


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


_REFINE_CODES = {'default': _ffi.REFINE_DEFAULT, 'float64': _ffi.REFINE_F64, 'float32': _ffi.REFINE_F32}


def xcorr_refine_batch(ref, img, upsample=1, cc_type='CC', return_status=False, refine='default'):
    """Shifts of ``img[k]`` relative to ``ref[k]`` for a batch of cutout pairs.

    ref, img : ``[N, ny, nx]`` float32 or float64, torch CUDA tensors (used in place) or
        numpy arrays (copied to the device).  5 <= ny, nx <= 682 (``_ffi.MAX_SIDE``; above 128 px the
        general path).  float64 pairs are masked
        and normalised in float64 (cc.py:131-156) before the float32 transforms.
    upsample : the cross-correlation is refined on a grid ``upsample`` times
        finer than the pixel grid before the 5x5 quadratic peak fit;
        ``upsample=2`` is the reference's half-pixel interlace (cc.py:121-126),
        ``upsample=1`` is ``fftconvolve(..,'same')`` + ``find_peak`` on one image.

    refine : ``'default'``, ``'float64'`` or ``'float32'`` -- the arithmetic of that refinement on cutouts of
        33..85 px per side (``SPX_REFINE_*``; up to 32 px it is float32, above 85 px float64 regardless).
        ``'default'`` is float32 matrix products.  float64 accumulation is 4-7x closer to the float64
        definition (64 px, upsample 10: 1.2e-5 instead of 5.5e-5 px) for 14 % (upsample 20: 21 %, upsample
        28 and above: 29..40 %) fewer pairs per second.  The distance grows with the width of the spot: for spots
        of sigma 11..15 px float32 loses no pair up to upsample 27 but 2 % / 10 % of them (beyond 1e-3 px) at
        upsample 39 / 59, float64 none -- ask for ``'float64'`` there; spots that fill their cutout (sigma beyond
        min(15 px, side / 6)) lose some pairs in both forms, float64 far fewer (``profiles/r03/refine_precision.txt``,
        ``width_precision_256.txt``, ``bench_64_u10_refine_f64.json``).

    Returns ``dxdy [N, 2]`` float64 (torch CUDA tensor if the inputs were
    tensors, else numpy) and, with ``return_status``, the int32 ``status [N]``.
    """
    if refine not in _REFINE_CODES:
        raise ValueError("refine must be 'default', 'float64' or 'float32'.")
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
        fn = lib.spx_xcorr_refine_ex_f64 if dt == torch.float64 else lib.spx_xcorr_refine_ex_f32
        _ffi.check(fn(
            device.ptr(r), device.ptr(m), n, ny, nx, int(upsample), _cc_code(cc_type),
            _REFINE_CODES[refine],
            device.ptr(out), device.ptr(status), device.ptr(ws), ws_bytes, device.stream_ptr()))
    if return_status:
        return _finish(out, like_torch), _finish(status, like_torch)
    return _finish(out, like_torch)


def find_displacement_batch(ref, im4, cc_type='NCC', full_output=False, return_status=False):
    """``find_displacement`` for a batch: ``ref [N, ny, nx]``, ``im4 [N, 4, ny, nx]``
    (image00, image10, image01, image11).  Returns ``dxdy [N, 2]`` float64, plus the
    interlaced images ``icc [N, 2ny, 2nx]`` with ``full_output`` and the status
    array with ``return_status``.  ``icc`` has the dtype the reference's has -- that of the cutouts
    (cc.py:121 ``np.empty(..., dtype=cc00.dtype)``): float64 for float64 cutouts, float32 otherwise; its
    values come from the float32 transforms either way (3e-6 relative to a float64 correlation)."""
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
        res.append(_finish(icc.to(dt), like_torch))          # cc.py:121: icc in the cutouts' dtype
    if return_status:
        res.append(_finish(status, like_torch))
    return res[0] if len(res) == 1 else tuple(res)


_FAMILIES = (32, 64, 85, 128)        # largest side of the kernel families that take mixed shapes


def find_displacement_var(refs, im4s, cc_type='NCC', full_output=False, return_status=False):
    """``find_displacement`` for cutouts of DIFFERENT shapes: ``refs[k]`` is a 2-D array, ``im4s[k]`` its
    four dithered blots (a ``[4, ny, nx]`` array or a 4-sequence, order 00, 10, 01, 11).  The reference's
    cutouts are bounding boxes + padding, one shape per source (cutout.py:159-175), and its loop
    (align.py:656-699) takes them one at a time; here all sources of one kernel family (sides up to 32 /
    64 / 85 / 128 px) go in ONE launch (``spx_find_displacement5_var_*``), larger ones in one launch per shape.

    Returns ``dxdy [N, 2]`` float64 (numpy), then the list of interlaced images with ``full_output``, then
    ``status [N]`` with ``return_status``; a cutout the kernels do not take (side below 3 or above 682) is
    not measured: (nan, nan), status -1."""
    n = len(refs)
    dxdy = np.full((n, 2), np.nan)
    status = np.full(n, -1, dtype=np.int32)
    iccs = [None] * n
    groups, big = {}, {}
    for k in range(n):
        r = np.asarray(refs[k])
        b = [np.asarray(x) for x in im4s[k]]
        if r.ndim != 2 or len(b) != 4 or any(x.shape != r.shape for x in b):
            raise ValueError("All cutouts must have same shape.")           # cc.py:103-105
        side = max(r.shape)
        if min(r.shape) < 3 or side > _ffi.MAX_SIDE:
            continue
        dt = np.float64 if r.dtype == np.float64 and all(x.dtype == np.float64 for x in b) else np.float32
        if side > _FAMILIES[-1]:
            big.setdefault((r.shape, dt), []).append(k)
        else:
            fam = next(f for f in _FAMILIES if side <= f)
            groups.setdefault((fam, dt), []).append(k)
    lib = _ffi.load()
    dev = device.init()
    for (fam, dt), idx in groups.items():
        shapes = np.array([np.shape(refs[k]) for k in idx], dtype=np.int32)
        sizes = shapes[:, 0].astype(np.int64) * shapes[:, 1]
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        total = int(sizes.sum())
        # packed back to back: [item][pixels] and [item][dither][pixels]
        ref_flat = np.concatenate([np.asarray(refs[k], dtype=dt).ravel() for k in idx])
        im4_flat = np.concatenate([np.asarray(x, dtype=dt).ravel() for k in idx for x in im4s[k]])
        tdt = torch.float64 if dt == np.float64 else torch.float32
        r_d = device.to_device(ref_flat, tdt)
        m_d = device.to_device(im4_flat, tdt)
        o_d = device.to_device(offs, torch.int64)
        s_d = device.to_device(shapes.ravel(), torch.int32)
        out = torch.empty((len(idx), 2), dtype=torch.float64, device=r_d.device)
        st = torch.empty((len(idx),), dtype=torch.int32, device=r_d.device)
        icc = torch.empty((4 * total,), dtype=torch.float32, device=r_d.device)
        fn = lib.spx_find_displacement5_var_f64 if dt == np.float64 else lib.spx_find_displacement5_var_f32
        with torch.cuda.device(r_d.device):
            ws, ws_bytes = _workspace(lib.spx_workspace_bytes_xcorr(len(idx), fam, fam), r_d.device)
            _ffi.check(fn(device.ptr(r_d), device.ptr(m_d), device.ptr(o_d), device.ptr(s_d), len(idx), fam,
                          _cc_code(cc_type), device.ptr(out), device.ptr(st), device.ptr(icc),
                          device.ptr(ws), ws_bytes, device.stream_ptr()))
        dxdy[idx] = out.cpu().numpy()
        status[idx] = st.cpu().numpy()
        if full_output:
            icc_h = icc.cpu().numpy()
            for j, k in enumerate(idx):
                o, (ny, nx) = int(offs[j]), shapes[j]
                iccs[k] = icc_h[4 * o:4 * o + 4 * ny * nx].reshape(2 * ny, 2 * nx).astype(dt, copy=False)
    for (shape, dt), idx in big.items():            # general path: one launch per shape
        ref = np.stack([np.asarray(refs[k], dtype=dt) for k in idx])
        im4 = np.stack([np.stack([np.asarray(x, dtype=dt) for x in im4s[k]]) for k in idx])
        d, icc, st = find_displacement_batch(ref, im4, cc_type=cc_type, full_output=True, return_status=True)
        dxdy[idx] = d
        status[idx] = st
        if full_output:
            for j, k in enumerate(idx):
                iccs[k] = icc[j]
    res = [dxdy]
    if full_output:
        res.append(iccs)
    if return_status:
        res.append(status)
    return res[0] if len(res) == 1 else tuple(res)


def find_displacement_packed(ref, im4, offsets, shapes, shapes_host, cc_type='NCC'):
    """``find_displacement`` for a whole catalog already packed on the device
    (``spx_find_displacement5_catalog_f32``): ``ref`` float32 [total] holds the reference cutouts back to
    back (item k = ``shapes[k] = (h, w)`` pixels at ``offsets[k]``), ``im4`` float32 [4 total] their four
    blots at ``4 offsets[k]``; ``offsets`` / ``shapes`` are CUDA tensors, ``shapes_host`` the same shapes as a
    numpy array (it only steers which kernel families are launched: no device round trip).

    Returns CUDA tensors ``(dxdy [N, 2] float64, status [N] int32, icc float32 [4 total])``; items no kernel
    family takes (a side below 3 or above 128 px) keep ``(nan, nan)`` / status -1."""
    n = int(offsets.shape[0])
    dev = ref.device
    out = torch.full((n, 2), float('nan'), dtype=torch.float64, device=dev)
    status = torch.full((n,), -1, dtype=torch.int32, device=dev)
    icc = torch.empty((max(int(im4.shape[0]), 1),), dtype=torch.float32, device=dev)
    side = np.asarray(shapes_host).max(axis=1) if n else np.zeros(0, int)
    low = np.asarray(shapes_host).min(axis=1) if n else np.zeros(0, int)
    ok = low >= 3
    mask = 0
    for bit, lo, hi in ((1, 2, 32), (2, 32, 64), (4, 64, 85), (8, 85, 128)):
        if np.any(ok & (side > lo) & (side <= hi)):
            mask |= bit
    lib = _ffi.load()
    with torch.cuda.device(dev):
        ws, ws_bytes = _workspace(lib.spx_workspace_bytes_xcorr(n, 128, 128) if mask & 8 else 0, dev)
        _ffi.check(lib.spx_find_displacement5_catalog_f32(
            device.ptr(ref), device.ptr(im4), device.ptr(offsets), device.ptr(shapes), n, mask,
            _cc_code(cc_type), device.ptr(out), device.ptr(status), device.ptr(icc), device.ptr(ws), ws_bytes,
            device.stream_ptr()))
    return out, status, icc


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
