#!/usr/bin/env python3
"""Static FLOP / instruction census of one kernel from hipcc's --save-temps ISA
(/tmp/isa/*.s, written by tools/kernel_regs.py): floating-point operations per WAVE by
instruction class, for DESIGN.md's compute-roofline figure (roofline.compute_fraction in
bench.py).  The per-pair body of the pair kernels is straight-line code (fully unrolled
transforms), executed once per pair by each of the workgroup's waves; the few real loops
(staging: 4 iterations, already unrolled; fine arg-max; refinement re-centring, normally one pass)
are counted once, which is what a pair normally executes.

    python tools/kernel_flops.py 'pair_kernel<2, 1, 0, false, float>' [waves_per_pair]
"""
import re
import subprocess
import sys

path = '/tmp/isa/spx_capi-hip-amdgcn-amd-amdhsa-gfx950.s'
want = sys.argv[1] if len(sys.argv) > 1 else 'pair_kernel<2, 1, 0, false, float>'
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 4
txt = open(path).read()
# kernel bodies: from "<mangled>:" label to its .Lfunc_end
names = re.findall(r'^(_ZN3spx\w+):', txt, flags=re.M)
sel = None
for m in names:
    dem = subprocess.run(['c++filt', m], capture_output=True, text=True).stdout.strip()
    if want in dem:
        sel = m
        break
assert sel, 'kernel not found: ' + want
body = txt[txt.index('\n' + sel + ':'):]
body = body[:body.index('.Lfunc_end')]
# FLOPs per lane per instruction
table = [
    (r'v_pk_fma_f32', 4), (r'v_pk_(add|mul)_f32', 2),
    (r'v_(fma|fmac|mad|mac)_f32', 2), (r'v_(add|sub|subrev|mul|max|min)_f32', 1),
    (r'v_(fma|fmac)_f64', 2), (r'v_(add|mul)_f64', 1),
]
counts = {}
valu = 0
for line in body.split('\n'):
    ins = line.strip().split(' ')[0]
    if ins.startswith('v_'):
        valu += 1
    counts[ins] = counts.get(ins, 0) + 1
flop_lane = 0
rows = []
for pat, f in table:
    n = sum(c for i, c in counts.items() if re.fullmatch(pat + r'(_e32|_e64|_dpp|_sdwa)?', i))
    rows.append((pat, n, f))
    flop_lane += n * f
mfma = sum(c for i, c in counts.items() if i.startswith('v_mfma_f32_16x16x4'))
mfma_flop_wave = mfma * 16 * 16 * 4 * 2
print('kernel', want)
for pat, n, f in rows:
    print('  %-40s %6d x %d flop/lane' % (pat, n, f))
print('  %-40s %6d x 2048 flop/wave' % ('v_mfma_f32_16x16x4_f32', mfma))
print('  VALU instructions (static) %d, LDS %d, global %d' % (
    valu, sum(c for i, c in counts.items() if i.startswith('ds_')),
    sum(c for i, c in counts.items() if i.startswith('global_'))))
vec = flop_lane * 64 * waves
mat = mfma_flop_wave * waves
print('  per pair (%d waves): vector %.3f MFLOP + matrix %.3f MFLOP = %.3f MFLOP' % (
    waves, vec / 1e6, mat / 1e6, (vec + mat) / 1e6))
