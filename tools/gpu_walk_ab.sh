#!/bin/bash
# A/B of the XCD-aware item walk (first_item) against the linear one, same box, alternating runs
set -o pipefail
mkdir -p gpurun_out/ab
C=subpixal_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function"
( /opt/rocm/bin/hipcc $F -o gpurun_out/ab/lib_xcd.so $C/spx_capi.hip > gpurun_out/ab/build_xcd.log 2>&1; echo "build xcd rc $?" ) &
( /opt/rocm/bin/hipcc $F -DSPX_LINEAR_WALK -o gpurun_out/ab/lib_linear.so $C/spx_capi.hip > gpurun_out/ab/build_linear.log 2>&1; echo "build linear rc $?" ) &
wait
for rep in 1 2; do
  for name in linear xcd; do
    cp gpurun_out/ab/lib_$name.so $C/libsubpixal_hip.so
    for cfg in "64 10" "80 10" "128 20" "32 10"; do set -- $cfg
      timeout -k 10 200 python bench.py --steps 30 --warmup 10 --tile $1 --upsample $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; [print('$name rep $rep tile $1  %.4g pairs/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
    done
  done
done 2>&1 | tee gpurun_out/ab/walk_ab.txt
rm -f gpurun_out/ab/lib_*.so
make -C subpixal_amd/csrc -B all > gpurun_out/ab/build_final.log 2>&1 && timeout -k 10 400 bash tools/gpu_pmc.sh 64 10 100000 > gpurun_out/ab/pmc64.log 2>&1; grep -E "read_bytes|write_bytes" gpurun_out/ab/pmc64.log
