"""Device plumbing: torch-ROCm provides device memory and streams, nothing else."""
import numpy as np
import torch

from . import _ffi

_initialised = set()


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("subpixal_amd needs an MI355X (no HIP device visible); "
                           "there is no CPU fallback.")


def init(device=None):
    """Select/initialise the HIP device used by this process (default: torch's
    current device).  Returns the device index."""
    require_gpu()
    lib = _ffi.load()
    if device is None:
        device = torch.cuda.current_device()
    device = int(torch.device('cuda', device).index if not isinstance(device, int) else device)
    if device not in _initialised:
        torch.cuda.set_device(device)
        _ffi.check(lib.spx_init(device))
        _initialised.add(device)
    return device


def prepare(upsample=1, shape=None):
    """Build the constant tables of ``upsample`` for every kernel family on the current device and
    raise the kernels' LDS limits (``spx_prepare``): afterwards launches neither allocate nor set
    function attributes, so they can be captured into a HIP graph.  Cutouts above 128 px (general
    path) need tables that depend on their size: pass ``shape=(ny, nx)`` (``spx_prepare_shape``)."""
    init()
    if shape is None:
        _ffi.check(_ffi.load().spx_prepare(int(upsample)))
    else:
        _ffi.check(_ffi.load().spx_prepare_shape(int(shape[0]), int(shape[1]), int(upsample)))


def shutdown():
    """Free the library's device tables (``spx_shutdown``); the next call rebuilds them."""
    _ffi.check(_ffi.load().spx_shutdown())
    _initialised.clear()


def to_device(a, dtype):
    """numpy array or torch tensor -> contiguous CUDA tensor of `dtype`."""
    dev = init()
    if isinstance(a, torch.Tensor):
        t = a
        if not t.is_cuda:
            t = t.to('cuda:%d' % dev)
        else:
            init(t.device.index)
        return t.to(dtype).contiguous()
    arr = np.ascontiguousarray(a)
    return torch.from_numpy(arr).to('cuda:%d' % dev).to(dtype).contiguous()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
