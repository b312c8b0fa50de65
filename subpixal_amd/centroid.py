"""Peak finding on the MI355X: drop-in for ``subpixal.centroid`` (reference:
/root/reference/subpixal/centroid.py:18-236)."""
import numpy as np
import torch

from . import _ffi, device

__all__ = ['find_peak', 'find_peak_batch']


def _process_box_pars(par):
    # centroid.py:239-253
    if hasattr(par, '__iter__'):
        if len(par) != 2:
            raise TypeError("Box specification must be either a scalar or "
                            "an iterable with two elements.")
        wx, wy = int(par[0]), int(par[1])
    else:
        wx = wy = int(par)
    if wx < 1 or wy < 1:
        raise ValueError("Box dimensions must be positive integer numbers.")
    return wx, wy


def find_peak_batch(images, guesses=None, peak_fit_box=5, peak_search_box=None, masks=None,
                    return_status=False):
    """``find_peak`` for ``images [N, ny, nx]`` (float64 on the device);
    ``guesses [N, 2]`` = (xmax, ymax) or None; ``masks [N, ny, nx]`` good-pixel
    booleans or None.  Returns ``xy [N, 2]`` float64."""
    like_torch = isinstance(images, torch.Tensor)
    img = device.to_device(images, torch.float64)
    if img.dim() != 3:
        raise ValueError("images must have shape [N, ny, nx].")
    n, ny, nx = img.shape
    if isinstance(peak_search_box, str):                     # centroid.py:102-109
        if peak_search_box == 'fitbox':
            peak_search_box = peak_fit_box
        elif peak_search_box == 'off':
            peak_search_box = None
        elif peak_search_box == 'all':
            peak_search_box = (ny, nx)                       # image_data.shape, as the reference
    g = None
    sbx = sby = 0
    if guesses is not None:
        g = device.to_device(guesses, torch.float64)
        if peak_search_box is not None:
            sbx, sby = _process_box_pars(peak_search_box)
    wx, wy = _process_box_pars(peak_fit_box)
    m = None
    if masks is not None:
        m = device.to_device(masks, torch.uint8)
        if tuple(m.shape) != tuple(img.shape):
            raise ValueError("mask must have the shape of the image.")
    out = torch.empty((n, 2), dtype=torch.float64, device=img.device)
    status = torch.empty((n,), dtype=torch.int32, device=img.device)
    lib = _ffi.load()
    with torch.cuda.device(img.device):
        _ffi.check(lib.spx_find_peak_f64(device.ptr(img), device.ptr(m), device.ptr(g), n, ny, nx,
                                         wx, wy, sbx, sby, device.ptr(out), device.ptr(status),
                                         device.stream_ptr()))
    if not like_torch:
        out, status = out.cpu().numpy(), status.cpu().numpy()
    return (out, status) if return_status else out


def find_peak(image_data, xmax=None, ymax=None, peak_fit_box=5,
              peak_search_box=None, mask=None):
    """Find location of the peak in an array by fitting a second degree 2D
    polynomial within ``peak_fit_box`` around the (optionally box-restricted,
    optionally masked) maximum pixel.

    Same signature, results and errors as the reference
    ``subpixal.centroid.find_peak`` (centroid.py:18-236), evaluated by the
    ``find_peak`` kernel of libsubpixal_hip.so in float64.
    """
    if (xmax is None) != (ymax is None):                      # centroid.py:93-96
        raise ValueError("Both 'xmax' and 'ymax' must be either None or not "
                         "None")
    image_data = np.asarray(image_data, dtype=np.float64)
    if mask is not None:
        mask = np.asarray(mask, dtype=bool)
        if xmax is None and not mask.any():
            raise ValueError("attempt to get argmax of an empty sequence")
    guesses = None if xmax is None else np.array([[xmax, ymax]], dtype=np.float64)
    xy = find_peak_batch(image_data[None], guesses, peak_fit_box, peak_search_box,
                         None if mask is None else mask[None])
    return (float(xy[0, 0]), float(xy[0, 1]))
