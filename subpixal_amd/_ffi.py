"""ctypes binding of libsubpixal_hip.so (include/subpixal_hip.h).

There is no CPU fallback: importing this module without the built library, or
calling into it without an MI355X visible, raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPX_HIP_LIB: another build of the same library (A/B measurements only: tools/gpu_r3_variants.sh)
LIB_PATH = os.environ.get('SPX_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libsubpixal_hip.so')

ABI_VERSION = 4
REFINE_DEFAULT, REFINE_F64, REFINE_F32 = 0, 1, 2         # SPX_REFINE_* (include/subpixal_hip.h)
MAX_SIDE = 682
MAX_UPSAMPLE = 59
MAX_UPSAMPLE_GENERAL = 39      # cutouts above 128 px

CC_CODES = {'CC': 0, 'NCC': 1, 'ZNCC': 2}

ST_OK, ST_EDGE, ST_NOMAX, ST_OUTSIDE, ST_WINDOW, ST_FEWPTS, ST_NONFINITE, ST_SHAPE = range(8)

_c = ctypes
_vp = _c.c_void_p
_SIGNATURES = {
    'spx_abi_version': (_c.c_int, []),
    'spx_device_count': (_c.c_int, []),
    'spx_init': (_c.c_int, [_c.c_int]),
    'spx_prepare': (_c.c_int, [_c.c_int]),
    'spx_prepare_shape': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'spx_shutdown': (_c.c_int, []),
    'spx_last_error': (_c.c_char_p, []),
    'spx_workspace_bytes_xcorr': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int]),
    'spx_workspace_bytes_displacement5': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    'spx_xcorr_refine_f32': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_int, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'spx_xcorr_refine_f64': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_int, _vp, _vp, _vp, _c.c_size_t, _vp]),
    # ... with the refine stage's arithmetic chosen per call (REFINE_DEFAULT / REFINE_F64)
    'spx_xcorr_refine_ex_f32': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _c.c_int, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'spx_xcorr_refine_ex_f64': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _c.c_int, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'spx_find_displacement5_f32': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int,
                                              _c.c_int, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'spx_find_displacement5_f64': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int,
                                              _c.c_int, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'spx_find_displacement5_var_f32': (_c.c_int, [_vp, _vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _vp,
                                                  _vp, _vp, _c.c_size_t, _vp]),
    'spx_find_displacement5_var_f64': (_c.c_int, [_vp, _vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _vp,
                                                  _vp, _vp, _c.c_size_t, _vp]),
    'spx_find_displacement5_catalog_f32': (_c.c_int, [_vp, _vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _vp,
                                                      _vp, _vp, _c.c_size_t, _vp]),
    'spx_gather_cutouts_var_f32': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _vp, _c.c_int64, _vp, _c.c_float,
                                              _vp, _vp, _vp, _vp]),
    'spx_blot4_var_f32': (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _vp, _c.c_int, _vp, _vp, _vp, _vp, _vp]),
    'spx_find_peak_f64': (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                     _c.c_int, _c.c_int, _c.c_int, _vp, _vp, _vp]),
    'spx_gather_cutouts_f32': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _vp, _c.c_int64,
                                          _c.c_int, _c.c_int, _c.c_float, _vp, _vp, _vp, _vp]),
    'spx_label_bboxes_i32': (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int32, _vp, _vp, _vp]),
    'spx_blot_affine4_f32': (_c.c_int, [_vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _vp, _c.c_int,
                                        _c.c_int, _vp, _vp]),
    'spx_blot_poly4_f32': (_c.c_int, [_vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _c.c_int, _vp, _c.c_int,
                                      _c.c_int, _vp, _vp]),
    'spx_gen_gaussian_pairs_f32': (_c.c_int, [_c.c_uint64, _c.c_int64, _c.c_int64, _c.c_int,
                                              _c.c_float, _c.c_float, _c.c_float, _vp, _vp,
                                              _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class SubpixalHipError(RuntimeError):
    """A call into libsubpixal_hip.so returned an error code."""


def load():
    """Load (once) and return the ctypes library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsubpixal_hip.so is not built (%s); run `python -c 'import "
                "__graft_entry__ as g; g.build()'` or `make -C subpixal_amd/csrc`. "
                "subpixal_amd has no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.spx_abi_version() != ABI_VERSION:
            raise ImportError("libsubpixal_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().spx_last_error()
        raise SubpixalHipError("libsubpixal_hip error %d: %s"
                               % (rc, msg.decode() if msg else ''))
