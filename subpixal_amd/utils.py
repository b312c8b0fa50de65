"""Small helpers with the reference's names (subpixal/utils.py)."""
import numpy as np

__all__ = ['py2round']


def py2round(x):
    """Round half away from zero, like Python 2's ``round`` (utils.py:144-161).
    Scalar or array.  (The GPU ``find_peak`` applies the same rule to its initial
    guess inside the kernel; this is the host-side utility of the same name.)"""
    if hasattr(x, '__iter__'):
        x = np.asarray(x)
        return np.where(x >= 0.0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(x.dtype, copy=False)
    return np.floor(x + 0.5) if x >= 0.0 else np.ceil(x - 0.5)
