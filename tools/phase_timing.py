#!/usr/bin/env python
"""Diagnostic: time the pair kernel (64x64, upsample=10) cut short after each phase.
Needs the diagnostic library (`make -C subpixal_amd/csrc diag`), whose phase variants are
separately compiled template instantiations (the product library carries none of them)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subpixal_amd import synth, device
N = int(os.environ.get('PAIRS', 100000))
ref, img, truth = synth.gaussian_pairs(N, 64)
lib = ctypes.CDLL(os.path.join(ROOT, 'subpixal_amd', 'csrc', 'libsubpixal_hip_diag.so'))
vp = ctypes.c_void_p
lib.spx_diag_pair_phase.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]
assert lib.spx_init(0) == 0
out = torch.empty((N, 2), dtype=torch.float64, device='cuda')
st = torch.empty((N,), dtype=torch.int32, device='cuda')
names = {1: 'stage', 2: 'fwd A ffts', 3: 'fwd A twiddle', 4: 'transpose 1', 5: 'fwd B ffts', 6: 'unpack+product',
         7: 'inv A', 8: 'transpose 2', 9: 'inv B ffts', 10: 'planes', 11: 'coarse argmax', 12: 'fine window (MFMA)',
         13: 'fine argmax', 0: 'full (fit + store)'}
def run(k):
    rc = lib.spx_diag_pair_phase(ref.data_ptr(), img.data_ptr(), N, 64, 64, k, out.data_ptr(), st.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
prev = 0.0
for k in list(range(1, 14)) + [0]:
    for _ in range(2):
        run(k)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run(k)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print('%2d %-22s cumulative %8.3f ms   delta %8.3f ms   (%.1f ns/pair)' % (k, names[k], ms, ms - prev, ms * 1e6 / N))
    prev = ms
print('check full variant vs truth:', float((out - truth).abs().max()))
