#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a bench.py command to the fingerprinted traffic files
bench.py quotes as roofline.traffic (pair kernel) and reference_mode.traffic (reference-mode kernel).
MI355X_MICROARCH.md (HBM): both counters are in KiB; FETCH_SIZE counts half the bytes of wide coalesced reads on
gfx950 (doubled here); WRITE_SIZE is exact.

usage: tools/pmc_reduce.py FETCH_DIR WRITE_DIR OUT_DIR --tile 64 --upsample 10 --pairs 100000 [--sources 20000]
"""
import argparse, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def counter(dirname, pattern, name):
    vals, kern = [], None
    for f in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if pattern in k and 'gen_pairs' not in k and r['Counter_Name'] == name:
                vals.append(float(r['Counter_Value']))
                kern = k.split('(')[0]
    return (sum(vals) / len(vals) if vals else None), len(vals), kern


def reduce(fetch_dir, write_dir, pattern, units, bytes_per_unit, unit_key, extra):
    f, nf, kern = counter(fetch_dir, pattern, 'FETCH_SIZE')
    w, nw, _ = counter(write_dir, pattern, 'WRITE_SIZE')
    if f is None or w is None:
        return None
    rd, wr = 2.0 * f * 1024.0, w * 1024.0
    out = {'kernel': kern, 'launches_averaged': [nf, nw], unit_key: units, 'kernel_build': bench.kernel_build(),
           'FETCH_SIZE_KB_per_launch': f, 'WRITE_SIZE_KB_per_launch': w, 'read_bytes_per_launch': rd,
           'write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr,
           'algorithmic_bytes_per_launch': units * bytes_per_unit, 'ratio': (rd + wr) / (units * bytes_per_unit),
           'note': '2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per launch; the counters sit on the L2-to-fabric side: '
                   'bytes served by the Infinity Cache count too'}
    out.update(extra)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir'); ap.add_argument('write_dir'); ap.add_argument('out_dir')
    ap.add_argument('--tile', type=int, default=64); ap.add_argument('--upsample', type=int, default=10)
    ap.add_argument('--pairs', type=int, default=100000); ap.add_argument('--sources', type=int, default=0)
    a = ap.parse_args()
    pat = 'pair32_kernel' if a.tile <= 32 else ('pair128_kernel' if a.tile > 85 else 'pair_kernel')
    cmd = 'rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 ' \
          '--no-cpu-baseline --tile %d --upsample %d (one pass per counter, tools/gpu_r3_evidence.sh)' % (a.tile, a.upsample)
    r = reduce(a.fetch_dir, a.write_dir, pat, a.pairs, 2 * a.tile * a.tile * 4 + 20, 'pairs_per_launch',
               {'tile': a.tile, 'upsample': a.upsample, 'command': cmd})
    if r:
        name = 'pmc_traffic.json' if (a.tile, a.upsample) == (64, 10) else 'pmc_traffic_%d_u%d.json' % (a.tile, a.upsample)
        json.dump(r, open(os.path.join(a.out_dir, name), 'w'), indent=1)
        print(name, 'traffic %.4g B per launch = %.3fx algorithmic' % (r['hbm_bytes_per_launch'], r['ratio']))
    if a.sources:
        # reference mode: 5 cutouts in, result out (the interlaced image is optional output: counted apart)
        r = reduce(a.fetch_dir, a.write_dir, 'disp5', a.sources, 5 * a.tile * a.tile * 4 + 20, 'sources_per_launch',
                   {'cutout': a.tile, 'cc_type': 'NCC', 'command': cmd,
                    'bytes_with_interlaced_image': a.sources * (5 * a.tile * a.tile * 4 + 16 * a.tile * a.tile + 20)})
        if r:
            r['ratio_with_interlaced_image'] = r['hbm_bytes_per_launch'] / r['bytes_with_interlaced_image']
            json.dump(r, open(os.path.join(a.out_dir, 'pmc_traffic_disp5_%d.json' % a.tile), 'w'), indent=1)
            print('pmc_traffic_disp5_%d.json traffic %.4g B per launch = %.3fx (5 cutouts + result), %.3fx with the interlaced image'
                  % (a.tile, r['hbm_bytes_per_launch'], r['ratio'], r['ratio_with_interlaced_image']))


if __name__ == '__main__':
    main()
