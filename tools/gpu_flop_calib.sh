#!/bin/bash
# Do the SQ_INSTS_VALU_{FMA,ADD,MUL}_F32 counters count a packed f32 instruction once or twice?  (tools/calib/)
set -o pipefail
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/build_ab/flop_counter_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $GRAFT_REPO_ROOT/tools/calib/flop_counter_calib.hip || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FLOPS_FP32 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/flop_calib -- $BIN > $GRAFT_REPO_ROOT/gpurun_out/r03/flop_calib.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/r03/flop_calib.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee gpurun_out/r03/flop_counter_calibration.txt
import csv, glob
f = glob.glob('gpurun_out/r03/flop_calib/**/*counter_collection.csv', recursive=True)[0]
rows = {}
for r in csv.DictReader(open(f)):
    rows.setdefault(r['Kernel_Name'].split('(')[0], {})[r['Counter_Name']] = float(r['Counter_Value'])
issued = 4096 * 1024
print('wave-instructions issued per kernel: %d' % issued)
for k, v in rows.items():
    print('%-12s' % k, '  '.join('%s %.3f' % (c.replace('SQ_INSTS_VALU', ''), v[c] / issued) for c in sorted(v)), '(counts per issued instruction)')
PY
