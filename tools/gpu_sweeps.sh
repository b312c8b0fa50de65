#!/bin/bash
# fresh-seed randomized sweeps (GPU vs oracle), reference mode and pair mode
set -o pipefail
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
rc=0
for seed in 9 21 33 47; do
  timeout -k 10 600 python tools/sweep_disp5.py --catalogs 600 --seed $seed > $O/sweep_disp5_seed$seed.txt 2>&1 || rc=1
  grep -E "sources in|MISMATCH|ill-cond" $O/sweep_disp5_seed$seed.txt
done
for seed in 11 23; do
  timeout -k 10 900 python tools/sweep_parity.py --trials 8000 --seed $seed > $O/sweep_seed$seed.txt 2>&1 || rc=1
  grep -E "trials in|MISMATCH" $O/sweep_seed$seed.txt
done
exit $rc
