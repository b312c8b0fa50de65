#!/usr/bin/env python3
"""Reduce the rocprofv3 SQ counter passes of tools/gpu_sq.sh to per-launch means of the bench kernel and, when the
FLOP-class counters exist, to gpurun_out/sq_flops_<tile>_u<U>_<tag>.json (what bench.py's compute_fraction reads).
usage: tools/sq_summary.py TAG [--tile 64 --upsample 10 --pairs 100000]"""
import argparse, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('tag')
ap.add_argument('--tile', type=int, default=64)
ap.add_argument('--upsample', type=int, default=10)
ap.add_argument('--pairs', type=int, default=100000)
a = ap.parse_args()
import bench
vals, kern = {}, None
for grp in ('sq', 'sq2', 'sqf'):
    for f in glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_%s_%s' % (grp, a.tag), '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'pair' in k and 'gen_pairs' not in k:
                kern = k.split('(')[0]
                vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in vals.items()}
waves = 8 if 'pair8' in (kern or '') else (1 if 'pair32' in (kern or '') else 4)
out = {'kernel': kern, 'tag': a.tag, 'pairs_per_launch': a.pairs, 'tile': a.tile, 'upsample': a.upsample,
       'kernel_build': bench.kernel_build(), 'waves_per_pair': waves, 'counters_per_launch': m}
if 'SQ_INSTS_VALU' in m:
    out['valu_insts_per_wave_pair'] = m['SQ_INSTS_VALU'] / (a.pairs * waves)
if 'SQ_ACTIVE_INST_VALU' in m and 'SQ_WAVE_CYCLES' in m:
    out['valu_active_fraction_of_wave_cycles'] = m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']
if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m:
    out['wait_any_fraction'] = m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']
    out['wait_inst_any_fraction'] = m.get('SQ_WAIT_INST_ANY', 0.0) / m['SQ_WAVE_CYCLES']
if 'SQ_INSTS_VALU_FLOPS_FP32' in m:
    # FLOPS_FP32/FP64: FLOP per lane per wave-instruction, packed instructions counted at 2 operations per lane
    # (calibrated, profiles/r03/flop_counter_calibration.txt) -> x 64 lanes; MFMA_MOPS: 512 FLOP each
    vec = 64.0 * (m.get('SQ_INSTS_VALU_FLOPS_FP32', 0) + m.get('SQ_INSTS_VALU_FLOPS_FP64', 0))
    mat = 512.0 * (m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) + m.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0))
    out.update({'vector_flop_per_pair': vec / a.pairs, 'matrix_flop_per_pair': mat / a.pairs,
                'flop_per_pair': (vec + mat) / a.pairs,
                'note': 'dynamic: 64 x (SQ_INSTS_VALU_FLOPS_FP32 + _FP64) + 512 x SQ_INSTS_VALU_MFMA_MOPS_F32/F64 of the bench '
                        'kernel, per launch / pairs (lanes switched off by EXEC are counted as if active)'})
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'sq_flops_%d_u%d_%s.json' % (a.tile, a.upsample, a.tag)), 'w'), indent=1)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'sq_summary_%s.json' % a.tag), 'w'), indent=1)
print(json.dumps(out, indent=1))
