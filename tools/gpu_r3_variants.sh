#!/bin/bash
# A/B of library variants built beforehand into build_ab/lib_<name>.so (they travel with the snapshot):
# alternating bench runs of the 64 tile on 4 and 8 waves per pair.   usage: tools/gpu_r3_variants.sh name...
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03/variants_$(echo "$@" | tr ' ' '_').txt
: > $O
for rep in 1 2 3; do
  for name in base "$@"; do
    lib=$PWD/build_ab/lib_$name.so; [ $name = base ] && lib=$PWD/subpixal_amd/csrc/libsubpixal_hip.so
    for w in ${WAVES:-4 8}; do
      SPX_HIP_LIB=$lib SPX_PAIR64_WAVES=$w timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-reference-mode 2>/dev/null | python -c "import sys,json; [print('$name waves $w rep $rep  %.4g pairs/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" | tee -a $O || exit 1
    done
  done
done
