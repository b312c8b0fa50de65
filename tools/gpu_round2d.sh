#!/bin/bash
# late round-2 pass: launch-shape sweep on the final kernels + fresh-seed randomized sweeps + soak
set -o pipefail
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
for g in 2 4 8 16 32 64 128; do
  SPX_GRID_PER_CU=$g timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; [print('SPX_GRID_PER_CU=%3d  %.4g pairs/s  kernel %.3f ms' % ($g, d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
done | tee $O/grid_sweep.txt
for g in 1 2 3; do
  SPX_GRID128_PER_CU=$g timeout -k 10 200 python bench.py --steps 20 --warmup 5 --tile 128 --upsample 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; [print('SPX_GRID128_PER_CU=%d  %.4g pairs/s  kernel %.3f ms' % ($g, d['value'], d['roofline']['kernel_ms'])) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" || exit 1
done | tee -a $O/grid_sweep.txt
timeout -k 10 900 python tools/sweep_parity.py --trials 6000 --seed 11 > $O/sweep_seed11.txt 2>&1 && tail -8 $O/sweep_seed11.txt
timeout -k 10 600 python tools/sweep_disp5.py --catalogs 300 --seed 9 > $O/sweep_disp5_seed9.txt 2>&1 && tail -6 $O/sweep_disp5_seed9.txt
timeout -k 10 600 python tools/soak_determinism.py > $O/soak.txt 2>&1 && tail -3 $O/soak.txt
