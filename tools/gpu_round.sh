#!/bin/bash
# One gpurun call: GPU tests, smoke, a bench run and a rocprofv3 kernel-trace of it.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi" > gpurun_out/run.log
(rocm-smi --showproductname 2>&1 | head -20) >> gpurun_out/run.log
echo "== pytest -m gpu" >> gpurun_out/run.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -25 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then echo "pytest killed/timed out (rc=$rc): stopping"; exit $rc; fi
echo "== smoke" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log &&
echo "== bench" && timeout -k 10 600 python bench.py --steps 10 --warmup 2 2>&1 | tee gpurun_out/bench.log | tail -5 &&
echo "== rocprof" && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1) &&
find gpurun_out/prof -name "*stats*" | head
