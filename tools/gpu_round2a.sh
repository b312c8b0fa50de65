#!/bin/bash
# round-2 measurement pass A: tests, benches for configs 2/3, shapes, 192-path stamps and PMC
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 200 python bench.py --steps 20 > gpurun_out/bench.log 2>&1 && tail -1 gpurun_out/bench.log | cut -c1-400
timeout -k 10 200 python bench.py --steps 10 --tile 128 --upsample 20 --no-cpu-baseline > gpurun_out/bench_config3.log 2>&1 && tail -1 gpurun_out/bench_config3.log | cut -c1-300
timeout -k 10 300 python tools/bench_shapes.py > gpurun_out/shapes.txt 2>&1; cat gpurun_out/shapes.txt
N=20000 timeout -k 10 200 python tools/phase_cycles128.py > gpurun_out/phase_cycles128.txt 2>&1; cat gpurun_out/phase_cycles128.txt
timeout -k 10 600 bash tools/gpu_pmc.sh 128 20 100000 > gpurun_out/pmc128.log 2>&1; tail -25 gpurun_out/pmc128.log
