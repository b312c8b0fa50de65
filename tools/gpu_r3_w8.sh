#!/bin/bash
# round 3, item 1: the eight-wave 64-tile kernel against the four-wave one on the same box:
# parity tests on the new kernel, alternating bench runs (SPX_PAIR64_WAVES = 4 | 8), phase stamps of both.
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity_w8.log 2>&1; rc=$?
tail -3 $O/pytest_parity_w8.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for w in 4 8; do
    SPX_PAIR64_WAVES=$w timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-reference-mode 2>/dev/null | python -c "import sys,json; [print('waves $w rep $rep  %.4g pairs/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
  done
done 2>&1 | tee $O/w8_ab.txt
for w in 4 8; do
  WAVES=$w timeout -k 10 300 python tools/phase_cycles.py 2>&1 | tee $O/phase_cycles64_w$w.txt || exit 1
done
