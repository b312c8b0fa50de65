#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== config 3"; timeout -k 10 300 python bench.py --tile 128 --upsample 20 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tee gpurun_out/bench_config3.log | python -c "import sys,json; [print({k:d[k] for k in ('value','ms_per_step')}) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" &&
echo "== 2-rank rehearsal (gloo, one device)" && timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --backend gloo --one-device --pairs 50000 2>&1 | tee gpurun_out/bench_2rank_rehearsal.log | tail -3 | cut -c1-400
