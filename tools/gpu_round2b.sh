#!/bin/bash
# quick loop for the period-192 kernel: parity tests above 85 px, stamps, config-3 bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "n128 or n96 or fold or nonfinite or float64 or shapes_cc or config3 or goldens or sweep" > gpurun_out/pytest_192.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_192.log
tail -4 gpurun_out/pytest_192.log
N=20000 timeout -k 10 200 python tools/phase_cycles128.py > gpurun_out/phase_cycles128.txt 2>&1; cat gpurun_out/phase_cycles128.txt
timeout -k 10 200 python bench.py --steps 10 --tile 128 --upsample 20 --no-cpu-baseline > gpurun_out/bench_config3.log 2>&1 && tail -1 gpurun_out/bench_config3.log | cut -c1-300
timeout -k 10 300 python tools/bench_shapes.py > gpurun_out/shapes.txt 2>&1; cat gpurun_out/shapes.txt
N=4000 timeout -k 10 300 python tools/bench_disp5.py 2>&1 | grep -v amdgpu | grep -E "n=(96|128|160)" 
