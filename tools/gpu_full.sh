#!/bin/bash
# full round: all gpu tests, smoke, bench (with cpu baseline), rocprof stats, PMC traffic
set -o pipefail
bash tools/gpu_round.sh || exit 1
bash tools/gpu_pmc.sh || exit 1
timeout -k 10 300 python tools/phase_cycles.py > gpurun_out/phase_cycles.log 2>&1
timeout -k 10 300 python bench.py --tile 128 --upsample 20 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_config3.log 2>&1
tail -1 gpurun_out/bench_config3.log | cut -c1-200
