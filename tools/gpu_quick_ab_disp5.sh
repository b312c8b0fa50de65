#!/bin/bash
# on-box A/B for reference mode: the library as pushed against a rebuild of the working tree
set -o pipefail
mkdir -p gpurun_out/ab
C=subpixal_amd/csrc
cp $C/libsubpixal_hip.so gpurun_out/ab/lib_old.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -o gpurun_out/ab/lib_new.so $C/spx_capi.hip > gpurun_out/ab/build_new.log 2>&1 || { tail -20 gpurun_out/ab/build_new.log; exit 1; }
for rep in 1 2; do
  for name in old new; do
    cp gpurun_out/ab/lib_$name.so $C/libsubpixal_hip.so
    echo "== $name rep $rep"
    SIZES="${SIZES:-32,64,80}" N=20000 timeout -k 10 300 python tools/bench_disp5.py 2>&1 | grep -v amdgpu
  done
done 2>&1 | tee gpurun_out/ab/quick_ab_disp5.txt
cp gpurun_out/ab/lib_new.so $C/libsubpixal_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_streams.py -m gpu -x -q 2>&1 | tail -2
rm -f gpurun_out/ab/lib_*.so
