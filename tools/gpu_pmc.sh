#!/bin/bash
# PMC passes for the bench kernel: FETCH_SIZE and WRITE_SIZE in separate runs (TCC slot
# limits), with --kernel-trace only (no sys/hip traces).
#   tools/gpu_pmc.sh [TILE [UPSAMPLE [PAIRS]]]     (default 64 10 100000)
# Writes gpurun_out/pmc_<TILE>_<COUNTER>/ and gpurun_out/pmc_traffic_<TILE>_u<UPSAMPLE>.json
set -o pipefail
TILE=${1:-64}; UPS=${2:-10}; PAIRS=${3:-100000}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TILE}_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --tile $TILE --upsample $UPS --pairs $PAIRS > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TILE}_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_${TILE}_$c.log; exit 1; }
done
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/pmc_${TILE}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_${TILE}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
KERN=pair_kernel; [ $TILE -gt 85 ] && KERN=pair128_kernel; [ $TILE -le 32 ] && KERN=pair32_kernel
python3 tools/pmc_traffic.py $F $W --kernel $KERN --pairs $PAIRS --tile $TILE --upsample $UPS > gpurun_out/pmc_traffic_${TILE}_u${UPS}.json && cat gpurun_out/pmc_traffic_${TILE}_u${UPS}.json
