#!/usr/bin/env python
"""Diagnostic: per-phase shader-clock cycles of the real pair kernel (64x64, U=10),
from in-kernel stamps (diagnostic library, DBG=100)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subpixal_amd import synth
N = int(os.environ.get('PAIRS', 100000))
ref, img, truth = synth.gaussian_pairs(N, 64)
lib = ctypes.CDLL(os.path.join(ROOT, 'subpixal_amd', 'csrc', 'libsubpixal_hip_diag.so'))
vp = ctypes.c_void_p
lib.spx_diag_pair_phase.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]
assert lib.spx_init(0) == 0
out = torch.empty((N, 2), dtype=torch.float64, device='cuda')
st = torch.zeros((N + 40,), dtype=torch.int32, device='cuda')
WAVES = int(os.environ.get('WAVES', 4))        # 4: pair_kernel; 8: w8::pair8_kernel (spx_kernels8.h)
names = ['stage+norm+balance', 'tile load (+barrier)', 'fwd A: pretw + ffts', 'fwd A: twiddle', 'transpose 1', 'fwd B ffts',
         'Z^2', 'inv A: ffts + twiddle', 'transpose 2', 'inv B ffts', 'planes write (+2 barriers)', 'coarse argmax',
         'fine window MFMA (+class sum)', 'fine argmax', 'store', 'end barrier', 'window decision', '5x5 fit (one wave)']
def run():
    rc = lib.spx_diag_pair_phase(ref.data_ptr(), img.data_ptr(), N, 64, 64, 100 if WAVES == 4 else 300, out.data_ptr(), st.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
run(); torch.cuda.synchronize()
st.zero_(); run(); torch.cuda.synchronize()
cyc = st[N:N + 40].cpu().numpy().view('uint64')
tot = cyc.sum()
nw = WAVES * N     # waves x pairs
if WAVES == 8:
    names[0], names[1], names[10] = 'fetch+balance+stage (2 barriers)', 'tile load + fold (+barrier)', 'recombine + exchange + planes (2 barriers)'
print('waves per pair', WAVES)
print('err vs truth', float((out - truth).abs().max()))
for i, nme in enumerate(names):
    print('%2d %-32s %9.0f cycles/pair/wave  %5.1f %%' % (i, nme, cyc[i] / nw, 100.0 * cyc[i] / tot))
print('total %.0f cycles per pair per wave' % (tot / nw))
