#!/bin/bash
# SQ counter passes of the bench command (one rocprofv3 --pmc run per counter group; --kernel-trace only).
#   tools/gpu_sq.sh [TAG]        env SPX_PAIR64_WAVES picks the 64-tile kernel; writes gpurun_out/pmc_sq*_TAG
set -o pipefail
TAG=${1:-w4}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-mode"
run() { # name counters...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_$TAG -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_$TAG.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_$TAG.log; return 1; }
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS || exit 1
# dynamic FLOP census.  SQ_INSTS_VALU_FLOPS_FP32 counts per wave-instruction 2 for a scalar FMA,
# 4 for v_pk_fma_f32, 2 for v_pk_add/mul_f32 (calibrated: tools/gpu_flop_calib.sh, profiles/r03/flop_counter_calibration.txt;
# the per-class counters FMA/ADD/MUL_F32 count a packed instruction ONCE), MFMA_MOPS 512 FLOP each
run sqf SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 || echo "FLOP counters not available on this build of rocprofv3"
cd $GRAFT_REPO_ROOT
python3 tools/sq_summary.py $TAG
