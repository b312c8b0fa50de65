#!/bin/bash
# round 3: full GPU test pass, then the headline / config-3 / reference-mode rates of the shipped library
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
python bench.py --no-cpu-baseline > $O/bench_64.json 2>/dev/null || exit 1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03/bench_64.json'))
print('64 tile: %.4g pairs/s, kernel %.3f ms; reference mode %.4g displacements/s' % (d['value'], d['roofline']['kernel_ms'], d['reference_mode']['value']))
PY
for cfg in "128 20" "96 10" "80 10" "32 10"; do set -- $cfg
  python bench.py --tile $1 --upsample $2 --no-cpu-baseline --no-reference-mode 2>/dev/null > $O/bench_$1_u$2.json || exit 1
  python -c "import json; d=json.load(open('$O/bench_$1_u$2.json')); print('tile $1 U $2: %.4g pairs/s, kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))"
done
SIZES=32,64,80,96,128 CC_TYPES=CC,NCC python tools/bench_disp5.py 2>/dev/null | tee $O/disp5.txt
