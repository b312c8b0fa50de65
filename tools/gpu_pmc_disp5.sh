#!/bin/bash
# HBM-side traffic of the reference-mode kernel (cc.find_displacement for a batch): FETCH_SIZE and
# WRITE_SIZE passes over tools/bench_disp5.py at one cutout size (default 64 px, NCC, 19968 sources)
set -o pipefail
N=${1:-64}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  SIZES=$N CC_TYPES=NCC N=20000 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_disp5_${N}_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_disp5.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_disp5_${N}_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_disp5_${N}_$c.log; exit 1; }
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = sorted(glob.glob('gpurun_out/pmc_disp5_${N}_%s/*/*counter_collection.csv' % c))[-1]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'disp5' in r['Kernel_Name'] and r['Counter_Name'] == c]
    out[c] = sum(v) / len(v)
    out['kernel'] = [r['Kernel_Name'] for r in csv.DictReader(open(f)) if 'disp5' in r['Kernel_Name']][0][:60]
n, count = $N, 19968
alg = count * (5 * n * n * 4 + 16 * n * n + 20)      # 5 cutouts in, interlaced image + result out
res = {'kernel': out['kernel'], 'sources_per_launch': count, 'cutout': n, 'cc_type': 'NCC',
       'read_bytes_per_launch': 2 * out['FETCH_SIZE'] * 1024, 'write_bytes_per_launch': out['WRITE_SIZE'] * 1024,
       'algorithmic_bytes_per_launch': alg}
res['hbm_bytes_per_launch'] = res['read_bytes_per_launch'] + res['write_bytes_per_launch']
res['ratio'] = res['hbm_bytes_per_launch'] / alg
json.dump(res, open('gpurun_out/pmc_traffic_disp5_${N}.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
