#!/bin/bash
# quick loop: gpu tests (pair-related) + bench + phase cycles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tee gpurun_out/bench_quick.log | python -c "import sys,json; [print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" &&
timeout -k 10 300 python tools/phase_cycles.py 2>&1 | tee gpurun_out/phase_cycles.log | tail -19
