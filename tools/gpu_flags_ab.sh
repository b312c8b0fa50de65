#!/bin/bash
# A/B of compiler scheduling options: build the library once per variant (in parallel, one hipcc
# process each), then bench each build on the three pair-mode families.
set -o pipefail
mkdir -p gpurun_out/ab
C=subpixal_amd/csrc
cp $C/libsubpixal_hip.so gpurun_out/ab/lib_shipped.so
# whatever happens below, the product library in the tree is the shipped build again when this script
# ends: a later step of the same lease (tests, bench, PMC passes) must never measure an A/B variant
restore() { [ -s gpurun_out/ab/lib_shipped.so ] && cp gpurun_out/ab/lib_shipped.so $C/libsubpixal_hip.so; rm -f gpurun_out/ab/lib_*.so; }
trap restore EXIT
names=(); i=0
while IFS='|' read -r name flags; do
  [ -z "$name" ] && continue
  names+=("$name")
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $flags \
      -o gpurun_out/ab/lib_$name.so $C/spx_capi.hip > gpurun_out/ab/build_$name.log 2>&1; echo "build $name rc $?" ) &
done <<'VARIANTS'
base|
maxilp|-mllvm -amdgpu-sched-strategy=max-ilp
maxclause|-mllvm -amdgpu-sched-strategy=max-memory-clause
bias0|-mllvm -amdgpu-schedule-metric-bias=0
bias100|-mllvm -amdgpu-schedule-metric-bias=100
trackers|-mllvm -amdgpu-use-amdgpu-trackers=1
nohighrp|-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1
VARIANTS
# progress lines while the builds run (a silent call is taken for hung)
while [ "$(jobs -r | wc -l)" -gt 0 ]; do sleep 45; echo "building: $(jobs -r | wc -l) left"; done
wait
for name in shipped "${names[@]}"; do
  [ -s gpurun_out/ab/lib_$name.so ] || { echo "== $name: no library"; continue; }
  cp gpurun_out/ab/lib_$name.so $C/libsubpixal_hip.so
  echo "== $name"
  for cfg in "64 10" "128 20" "80 10" "32 10"; do set -- $cfg
    timeout -k 10 200 python bench.py --steps 30 --warmup 10 --tile $1 --upsample $2 --no-cpu-baseline --no-reference-mode 2>/dev/null | python -c "import sys,json; [print('   tile $1 U $2  %.4g pairs/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
  done
done 2>&1 | tee gpurun_out/ab/results.txt
