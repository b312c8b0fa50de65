#!/bin/bash
# SQ / cache counters for the period-192 kernel (config 3): where its waves wait
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --tile 128 --upsample 20"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc128_$name -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc128_$name.log 2>&1 || { tail -5 $R/gpurun_out/pmc128_$name.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
echo done
cd $R
python - <<'PY'
import csv, glob, os
for name in ('sq1', 'sq2', 'tcc', 'tcp'):
    files = glob.glob('gpurun_out/pmc128_%s/*/*_counter_collection.csv' % name)
    if not files:
        print(name, 'no output'); continue
    agg = {}
    for r in csv.DictReader(open(sorted(files)[-1])):
        if 'pair128_kernel' in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('%-28s %.4g per launch (%d launches)' % (k, sum(v) / len(v), len(v)))
PY
