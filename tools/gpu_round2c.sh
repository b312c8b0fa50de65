#!/bin/bash
# quick loop for the 64-tile kernel: parity, stamps, bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py -m gpu -q -x > gpurun_out/pytest_64.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_64.log
tail -4 gpurun_out/pytest_64.log
timeout -k 10 200 python tools/phase_cycles.py 2>&1 | grep -v amdgpu > gpurun_out/phase_cycles.txt; cat gpurun_out/phase_cycles.txt
timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1 && tail -1 gpurun_out/bench_quick.log | cut -c1-200
timeout -k 10 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu > gpurun_out/shapes.txt; head -9 gpurun_out/shapes.txt
