#!/usr/bin/env python3
"""Per-launch means of the SQ counters of ONE kernel (name substring) from rocprofv3 --pmc output directories.
usage: tools/sq_kernel.py NAME_SUBSTRING UNITS_PER_LAUNCH DIR [DIR...]   (4 waves per unit assumed: one workgroup of
256 threads works on one unit at a time)"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
name, units = sys.argv[1], int(sys.argv[2])
vals, kern = {}, None
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r['Kernel_Name']:
                kern = r['Kernel_Name'].split('(')[0]
                vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in vals.items()}
out = {'kernel': kern, 'units_per_launch': units, 'kernel_build': bench.kernel_build(), 'launches_averaged': {k: len(v) for k, v in vals.items()},
       'counters_per_launch': m, 'counters_per_unit': {k: v / units for k, v in m.items()}}
if 'SQ_ACTIVE_INST_VALU' in m and 'SQ_WAVE_CYCLES' in m:
    out['valu_active_fraction_of_wave_cycles'] = m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']
    out['active_any_fraction'] = m.get('SQ_ACTIVE_INST_ANY', 0.0) / m['SQ_WAVE_CYCLES']
    out['wait_any_fraction'] = m.get('SQ_WAIT_ANY', 0.0) / m['SQ_WAVE_CYCLES']
    out['wait_inst_any_fraction'] = m.get('SQ_WAIT_INST_ANY', 0.0) / m['SQ_WAVE_CYCLES']
if 'SQ_INSTS_VALU' in m:
    out['valu_insts_per_wave_unit'] = m['SQ_INSTS_VALU'] / (units * 4)
print(json.dumps(out, indent=1))
