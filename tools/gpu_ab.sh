#!/bin/bash
# A/B: build the library with two launch-bound settings on the box and bench each.
set -o pipefail
mkdir -p gpurun_out
for lb in "nthreads" "nthreads, 2"; do
  sed -i "s/#define SPX_TKERNEL(nthreads) __global__ __launch_bounds__(.*)/#define SPX_TKERNEL(nthreads) __global__ __launch_bounds__($lb)/" subpixal_amd/csrc/spx_rt_hip.h
  make -C subpixal_amd/csrc -B all > gpurun_out/build_ab.log 2>&1 || exit 1
  echo "== launch_bounds($lb)" | tee -a gpurun_out/ab.log
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tee -a gpurun_out/ab.log | python -c "import sys,json; [print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
done
