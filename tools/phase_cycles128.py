#!/usr/bin/env python
"""In-kernel s_memtime stamps of the 128-tile pair kernel (diagnostic build, upsample 20):
shader cycles per pair per wave by phase."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from subpixal_amd import device, synth      # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'subpixal_amd', 'csrc', 'libsubpixal_hip_diag.so'))
vp = ctypes.c_void_p
lib.spx_init.argtypes = [ctypes.c_int]
lib.spx_workspace_bytes_xcorr.restype = ctypes.c_size_t
lib.spx_workspace_bytes_xcorr.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
lib.spx_diag_pair128_phase.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                       ctypes.c_size_t, vp]
device.init()
assert lib.spx_init(0) == 0
N = int(os.environ.get('N', 20000))
ref, img, truth = synth.gaussian_pairs(N, 128)
out = torch.zeros((N, 2), dtype=torch.float64, device='cuda')
st = torch.zeros((N + 64,), dtype=torch.int32, device='cuda')
nws = lib.spx_workspace_bytes_xcorr(N, 128, 128)
ws = torch.empty((nws,), dtype=torch.uint8, device='cuda')
names = ['norm + balance', 'stage y-fold + x-fold (3 rounds)', 'class FFTs', 'class planes -> workspace',
         'sync + combine + arg-max + sync', 'block arg-max', 'fine window MFMA', 'fine argmax', 'fit + store', 'end barrier']


def run():
    rc = lib.spx_diag_pair128_phase(ref.data_ptr(), img.data_ptr(), N, 128, 128, out.data_ptr(), st.data_ptr(),
                                    ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


run(); torch.cuda.synchronize()
st.zero_(); run(); torch.cuda.synchronize()
cyc = st[N:N + 40].cpu().numpy().view('uint64')
tot = cyc.sum()
nw = 4 * N
print('err vs truth', float((out - truth).abs().max()))
for i, nme in enumerate(names):
    print('%2d %-38s %9.0f cycles/pair/wave  %5.1f %%' % (i, nme, cyc[i] / nw, 100.0 * cyc[i] / tot))
print('total %.0f cycles per pair per wave' % (tot / nw))
