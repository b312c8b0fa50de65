"""Synthetic Gaussian-spot workloads, generated on the device."""
import numpy as np
import torch

from . import _ffi, device

__all__ = ['gaussian_pairs', 'pair_params']

_M64 = (1 << 64) - 1


def _splitmix(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    r = z
    r = ((r ^ (r >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    r = ((r ^ (r >> 27)) * 0x94D049BB133111EB) & _M64
    return z, r ^ (r >> 31)


def pair_params(seed, first_index, count, sigma_lo, sigma_hi, max_shift):
    """Host mirror of the device generator's parameter stream: returns
    ``tx, ty, sigma, amp`` (float64 arrays) for pairs first_index..+count."""
    out = np.empty((count, 4))
    for i in range(count):
        z = (seed + (first_index + i + 1) * 0xD1342543DE82EF95) & _M64
        u = []
        for _ in range(4):
            z, r = _splitmix(z)
            u.append((r >> 11) * (1.0 / 9007199254740992.0))
        out[i] = ((2 * u[0] - 1) * max_shift, (2 * u[1] - 1) * max_shift,
                  np.float32(sigma_lo) + u[2] * float(np.float32(sigma_hi) - np.float32(sigma_lo)),
                  0.5 + 1.5 * u[3])
    return out[:, 0], out[:, 1], out[:, 2], out[:, 3]


def gaussian_pairs(count, n, seed=20261003, first_index=0, sigma_lo=4.0, sigma_hi=6.0,
                   max_shift=3.0, dev=None):
    """``ref [count, n, n]``, ``img [count, n, n]`` float32 and ``truth [count, 2]``
    float64 CUDA tensors: ref has a Gaussian spot at the tile centre, img the same
    spot displaced by (tx, ty) ~ U(-max_shift, max_shift)."""
    d = device.init(dev)
    ref = torch.empty((count, n, n), dtype=torch.float32, device='cuda:%d' % d)
    img = torch.empty_like(ref)
    truth = torch.empty((count, 2), dtype=torch.float64, device=ref.device)
    lib = _ffi.load()
    with torch.cuda.device(ref.device):
        _ffi.check(lib.spx_gen_gaussian_pairs_f32(
            int(seed) & _M64, int(first_index), int(count), int(n), float(sigma_lo),
            float(sigma_hi), float(max_shift), device.ptr(ref), device.ptr(img),
            device.ptr(truth), device.stream_ptr()))
    return ref, img, truth
