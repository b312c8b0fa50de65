#!/usr/bin/env python3
"""Static FLOP / instruction census of kernels from hipcc's --save-temps ISA
(/tmp/isa/*.s, written by tools/kernel_regs.py): floating-point operations per WAVE by
instruction class, for DESIGN.md's compute-roofline figure (roofline.compute_fraction in
bench.py).  The per-pair body of the pair kernels is straight-line code (fully unrolled
transforms), executed once per pair by each of the workgroup's waves; the few real loops
(staging: 4 iterations, already unrolled; fine arg-max; refinement re-centring, normally one pass)
are counted once, which is what a pair normally executes.  STATIC: code of branches a pair does not
take (ragged staging instantiations, normalisation) is counted too -- an over-count; the dynamic
figure comes from the SQ_INSTS_VALU_* counter passes (tools/gpu_sq_flops.sh).

    python tools/kernel_flops.py 'pair_kernel<2, 1, 0, false, float, spx::RefineF32>' [waves_per_pair]
    python tools/kernel_flops.py --json OUT.json      # the table bench.py reads (all bench kernels)
"""
import json
import os
import re
import subprocess
import sys

ISA = '/tmp/isa/spx_capi-hip-amdgcn-amd-amdhsa-gfx950.s'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# FLOPs per lane per instruction
TABLE = [
    (r'v_pk_fma_f32', 4), (r'v_pk_(add|mul)_f32', 2),
    (r'v_(fma|fmac|mad|mac)_f32', 2), (r'v_(add|sub|subrev|mul|max|min)_f32', 1),
    (r'v_(fma|fmac)_f64', 2), (r'v_(add|mul)_f64', 1),
]


def demangled_index(txt):
    names = re.findall(r'^(_ZN3spx\w+):', txt, flags=re.M)
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
    return list(zip(names, out))


def census(txt, index, want, waves):
    sel = None
    for m, dem in index:
        if want in dem:
            sel = m
            break
    if sel is None:
        return None
    body = txt[txt.index('\n' + sel + ':'):]
    body = body[:body.index('.Lfunc_end')]
    counts = {}
    valu = 0
    for line in body.split('\n'):
        ins = line.strip().split(' ')[0]
        if ins.startswith('v_'):
            valu += 1
        counts[ins] = counts.get(ins, 0) + 1
    flop_lane = 0
    rows = []
    for pat, f in TABLE:
        n = sum(c for i, c in counts.items() if re.fullmatch(pat + r'(_e32|_e64|_dpp|_sdwa)?', i))
        rows.append((pat, n, f))
        flop_lane += n * f
    mfma32 = sum(c for i, c in counts.items() if i.startswith('v_mfma_f32_16x16x4'))
    mfma64 = sum(c for i, c in counts.items() if i.startswith('v_mfma_f64_16x16x4'))
    return {
        'kernel': want, 'waves_per_unit': waves, 'rows': rows, 'mfma_f32': mfma32, 'mfma_f64': mfma64,
        'valu_static': valu,
        'lds_static': sum(c for i, c in counts.items() if i.startswith('ds_')),
        'global_static': sum(c for i, c in counts.items() if i.startswith('global_')),
        'vector_mflop': flop_lane * 64 * waves / 1e6,
        'matrix_mflop': (mfma32 + mfma64) * 2048 * waves / 1e6,
    }


# the kernel instances bench.py can time: (family key, WB) -> (instance name, waves per pair)
BENCH_KERNELS = {
    '32:1': ('pair32_kernel<1, float>', 1),
    '64:1': ('pair_kernel<2, 1, 0, false, float, spx::RefineF32>', 4),
    '64fold:1': ('pair_kernel<2, 1, 0, true, float, spx::RefineF32>', 4),
    '64w8:1': ('pair8_kernel<1, 0, float>', 8),
}


def main():
    if not os.path.exists(ISA):
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'kernel_regs.py')],
                              stdout=subprocess.DEVNULL)
    txt = open(ISA).read()
    index = demangled_index(txt)
    if len(sys.argv) > 2 and sys.argv[1] == '--json':
        sys.path.insert(0, ROOT)
        import bench
        out = {'kernel_build': bench.kernel_build(),
               'method': 'static ISA census (tools/kernel_flops.py): FLOPs of every floating-point instruction in '
                         'the kernel instance x 64 lanes x waves per pair; over-counts branches a pair does not take',
               'kernels': {}}
        for key, (name, waves) in BENCH_KERNELS.items():
            c = census(txt, index, name, waves)
            if c is not None:
                out['kernels'][key] = {k: c[k] for k in ('kernel', 'waves_per_unit', 'vector_mflop', 'matrix_mflop',
                                                         'valu_static', 'mfma_f32', 'mfma_f64')}
        json.dump(out, open(sys.argv[2], 'w'), indent=1)
        print(json.dumps(out, indent=1))
        return
    want = sys.argv[1] if len(sys.argv) > 1 else 'pair_kernel<2, 1, 0, false, float, spx::RefineF32>'
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    c = census(txt, index, want, waves)
    assert c, 'kernel not found: ' + want
    print('kernel', want)
    for pat, n, f in c['rows']:
        print('  %-40s %6d x %d flop/lane' % (pat, n, f))
    print('  %-40s %6d x 2048 flop/wave' % ('v_mfma_f32_16x16x4_f32', c['mfma_f32']))
    print('  %-40s %6d x 2048 flop/wave' % ('v_mfma_f64_16x16x4_f64', c['mfma_f64']))
    print('  VALU instructions (static) %d, LDS %d, global %d' % (c['valu_static'], c['lds_static'], c['global_static']))
    print('  per pair (%d waves): vector %.3f MFLOP + matrix %.3f MFLOP = %.3f MFLOP' % (
        waves, c['vector_mflop'], c['matrix_mflop'], c['vector_mflop'] + c['matrix_mflop']))


if __name__ == '__main__':
    main()
