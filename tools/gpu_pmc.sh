#!/bin/bash
# PMC passes for the bench kernel: FETCH_SIZE and WRITE_SIZE in separate runs (TCC slot
# limits), with --kernel-trace only (no sys/hip traces).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log; exit 1; }
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc_* -name "*counter_collection*" | head
