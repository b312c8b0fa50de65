#!/bin/bash
# SQ counter passes of the bench command WITH its reference-mode block (tools/gpu_sq.sh switches that block off), reduced
# for the five-transform reference-mode kernel spx::p5::disp5p_kernel (20000 sources of 64x64 per launch, NCC).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { local name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_ref -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_ref.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_ref.log; return 1; }
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/sq_kernel.py disp5p_kernel 20000 gpurun_out/pmc_sq_ref gpurun_out/pmc_sq2_ref > gpurun_out/sq_summary_disp5p.json && cat gpurun_out/sq_summary_disp5p.json
