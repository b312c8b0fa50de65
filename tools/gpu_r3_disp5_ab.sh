#!/bin/bash
# reference mode on the 64 tile: the five-transform kernel (SPX_DISP5_PACKED=1, spx_kernels5.h) against round 2's
set -o pipefail
mkdir -p gpurun_out/r03
for rep in 1 2; do for p in 0 1; do
  SPX_DISP5_PACKED=$p SIZES=${SIZES:-40,64,80} CC_TYPES=${CC_TYPES:-CC,NCC} python tools/bench_disp5.py 2>/dev/null | sed "s/^/packed=$p rep $rep  /"
done; done | tee gpurun_out/r03/disp5_packed_ab.txt
