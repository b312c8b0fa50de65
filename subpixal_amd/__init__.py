"""subpixal_amd -- MI355X-native (gfx950) cross-correlation + sub-pixel peak
refinement for subpixal-style image alignment.

Module and function names follow the reference package (``subpixal.cc``,
``subpixal.centroid``, ``subpixal.align``, ``subpixal.cutout``, ``subpixal.utils``)
for the one hot path this package replaces; see DESIGN.md.
"""
from . import _ffi                                           # noqa: F401
from .cc import find_displacement, find_displacement_batch, find_displacement_var, xcorr_refine_batch
from .centroid import find_peak, find_peak_batch
from .utils import py2round

__version__ = '0.1.0'
__all__ = ['find_displacement', 'find_displacement_batch', 'find_displacement_var', 'xcorr_refine_batch',
           'find_peak', 'find_peak_batch', 'py2round']
