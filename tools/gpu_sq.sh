#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2.log; exit 1; }
echo done
