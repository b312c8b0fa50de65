#!/bin/bash
# round-2 evidence pass: bench + rocprofv3 kernel stats + PMC traffic + SQ counters + auxiliary rates
set -o pipefail
mkdir -p gpurun_out/r02
O=gpurun_out/r02
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $O/bench.json 2>$O/bench.err && cut -c1-200 $O/bench.json
for cfg in "128 20" "96 10" "80 10" "32 10"; do set -- $cfg
  timeout -k 10 200 python bench.py --steps 10 --tile $1 --upsample $2 --no-cpu-baseline > $O/bench_${1}_u$2.json 2>>$O/bench.err && cut -c1-160 $O/bench_${1}_u$2.json
done
echo "== rocprofv3 kernel stats (64 tile, then config 3)"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof64 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/bench_under_rocprofv3.json 2>$R/$O/rocprof64.err) && cut -c1-160 $O/bench_under_rocprofv3.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof128 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --tile 128 --upsample 20 > $R/$O/bench_128_under_rocprofv3.json 2>$R/$O/rocprof128.err) && cut -c1-160 $O/bench_128_under_rocprofv3.json
find $O/prof64 $O/prof128 -name "*kernel_stats.csv" | head
echo "== PMC traffic"
for cfg in "64 10" "128 20" "80 10" "96 10"; do set -- $cfg
  timeout -k 10 400 bash tools/gpu_pmc.sh $1 $2 100000 > $O/pmc_$1.log 2>&1; grep -E "hbm_bytes|algorithmic" $O/pmc_$1.log
done
timeout -k 10 400 bash tools/gpu_pmc_disp5.sh 64 > $O/pmc_disp5_64.log 2>&1; grep -E "ratio|hbm_bytes" $O/pmc_disp5_64.log
echo "== SQ counters (64 tile)"
timeout -k 10 600 bash tools/gpu_sq.sh > $O/sq.log 2>&1; tail -2 $O/sq.log
# (the traffic files just measured on this build, where bench.py looks for them)
cp gpurun_out/pmc_traffic_64_u10.json profiles/r02/pmc_traffic.json; for t in "128_u20" "80_u10" "96_u10"; do cp gpurun_out/pmc_traffic_$t.json profiles/r02/; done
timeout -k 10 300 python bench.py > $O/bench.json 2>$O/bench.err && cut -c1-200 $O/bench.json
timeout -k 10 200 python bench.py --tile 128 --upsample 20 --no-cpu-baseline > $O/bench_128_u20.json 2>>$O/bench.err
for cfg in "96 10" "80 10" "32 10"; do set -- $cfg
  timeout -k 10 200 python bench.py --tile $1 --upsample $2 --no-cpu-baseline > $O/bench_${1}_u$2.json 2>>$O/bench.err && cut -c1-160 $O/bench_${1}_u$2.json
done
echo "== auxiliary"
timeout -k 10 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu > $O/shapes.txt; cat $O/shapes.txt
N=20000 timeout -k 10 300 python tools/bench_disp5.py 2>&1 | grep -v amdgpu > $O/disp5.txt; cat $O/disp5.txt
timeout -k 10 300 python tools/bench_aux.py 2>&1 | grep -v amdgpu > $O/aux.txt; tail -8 $O/aux.txt
timeout -k 10 200 python tools/phase_cycles.py 2>&1 | grep -v amdgpu > $O/phase_cycles64.txt
N=20000 timeout -k 10 200 python tools/phase_cycles128.py 2>&1 | grep -v amdgpu > $O/phase_cycles128.txt
for u in 2 10 16 27 30 43 50; do timeout -k 10 100 python bench.py --steps 5 --upsample $u --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('upsample %2d  %.3e pairs/s' % ($u, d['value']))"; done > $O/usweep.txt; cat $O/usweep.txt
