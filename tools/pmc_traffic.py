#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc.sh) to
profiles/<round>/pmc_traffic.json, the HBM bytes per launch that bench.py reports as
roofline.traffic.  MI355X_MICROARCH.md (HBM section): both counters are in KiB; the guide's
gfx950 correction doubles FETCH_SIZE for wide coalesced reads; WRITE_SIZE is exact.

usage: tools/pmc_traffic.py FETCH.csv WRITE.csv [--kernel pair_kernel] [--pairs 100000] [--build TEXT] > out.json
"""
import argparse
import csv
import json


def mean_counter(path, kernel, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row['Kernel_Name'] and row['Counter_Name'] == counter:
                vals.append(float(row['Counter_Value']))
    if not vals:
        raise SystemExit(f'no {counter} rows for {kernel} in {path}')
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv')
    ap.add_argument('write_csv')
    ap.add_argument('--kernel', default='pair_kernel')
    ap.add_argument('--pairs', type=int, default=100000)
    ap.add_argument('--bytes-per-pair', type=int, default=2 * 64 * 64 * 4 + 20)
    ap.add_argument('--build', default='')
    ap.add_argument('--tile', type=int, default=64)
    ap.add_argument('--upsample', type=int, default=10)
    a = ap.parse_args()
    if a.bytes_per_pair == 2 * 64 * 64 * 4 + 20:
        a.bytes_per_pair = 2 * a.tile * a.tile * 4 + 20
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fetch_kb, nf = mean_counter(a.fetch_csv, a.kernel, 'FETCH_SIZE')
    write_kb, nw = mean_counter(a.write_csv, a.kernel, 'WRITE_SIZE')
    rd = 2.0 * fetch_kb * 1024.0
    wr = write_kb * 1024.0
    print(json.dumps({
        'command': 'rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py '
                   '--steps 3 --warmup 1 --no-cpu-baseline --tile %d --upsample %d (one pass per counter, '
                   'tools/gpu_pmc.sh)' % (a.tile, a.upsample),
        'kernel': a.kernel,
        'launches_averaged': [nf, nw],
        'pairs_per_launch': a.pairs,
        'tile': a.tile,
        'upsample': a.upsample,
        'build': a.build,
        'kernel_build': bench.kernel_build(),
        'FETCH_SIZE_KB_per_launch': fetch_kb,
        'WRITE_SIZE_KB_per_launch': write_kb,
        'read_bytes_per_launch': rd,
        'write_bytes_per_launch': wr,
        'hbm_bytes_per_launch': rd + wr,
        'algorithmic_bytes_per_launch': a.pairs * a.bytes_per_pair,
        'note': 'MI355X_MICROARCH.md (HBM): FETCH_SIZE counts half the bytes of wide coalesced reads on '
                'gfx950 (doubled here); WRITE_SIZE exact.  Excess over the algorithmic bytes = constant '
                'tables and scratch traffic of spilled VGPRs, if the build has any.',
    }, indent=1))


if __name__ == '__main__':
    main()
