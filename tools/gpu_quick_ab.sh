#!/bin/bash
# quick on-box A/B: the committed library (as pushed) against a rebuild of the working tree
set -o pipefail
mkdir -p gpurun_out/ab
C=subpixal_amd/csrc
cp $C/libsubpixal_hip.so gpurun_out/ab/lib_old.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -o gpurun_out/ab/lib_new.so $C/spx_capi.hip > gpurun_out/ab/build_new.log 2>&1 || { tail -20 gpurun_out/ab/build_new.log; exit 1; }
for rep in 1 2 3; do
  for name in old new; do
    cp gpurun_out/ab/lib_$name.so $C/libsubpixal_hip.so
    for cfg in "32 10" "24 10" "64 10"; do set -- $cfg
      timeout -k 10 200 python bench.py --steps 30 --warmup 10 --tile $1 --upsample $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; [print('$name rep $rep tile $1  %.4g pairs/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
    done
  done
done 2>&1 | tee gpurun_out/ab/quick_ab.txt
cp gpurun_out/ab/lib_new.so $C/libsubpixal_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
rm -f gpurun_out/ab/lib_*.so
