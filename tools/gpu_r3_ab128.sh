#!/bin/bash
# A/B on the period-192 path: the shipped library against build_ab/lib_<name>.so (alternating runs)
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab128_$(echo "$@" | tr ' ' '_').txt
: > $O
for rep in 1 2 3; do
  for name in base "$@"; do
    lib=$PWD/build_ab/lib_$name.so; [ $name = base ] && lib=$PWD/subpixal_amd/csrc/libsubpixal_hip.so
    for cfg in "128 20" "96 10"; do set -- $cfg
      SPX_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --tile $1 --upsample $2 --no-cpu-baseline --no-reference-mode 2>/dev/null | python -c "import sys,json; [print('$name tile $1 U $2 rep $rep  %.4g pairs/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" | tee -a $O || exit 1
    done
  done
done
