#!/bin/bash
# Round-3 evidence pass on the shipped library (run LAST: the traffic / FLOP files carry the kernel-source fingerprint).
# Writes gpurun_out/r03e/...; copy to profiles/r03/ afterwards (tools/collect_r03.sh).
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python3 $GRAFT_REPO_ROOT/bench.py"
step() { echo "== $1"; }
step "bench (default flags)"; $B > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
step "kernel trace"; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2> $O/trace.log || { tail -5 $O/trace.log; exit 1; }
for cfg in "64 10 20000" "128 20 0" "80 10 0"; do set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    step "pmc $c tile $1"
    X=""; [ $3 = 0 ] && X="--no-reference-mode"
    timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$1_$c -- $B --steps 3 --warmup 1 --no-cpu-baseline --tile $1 --upsample $2 $X > $O/pmc_$1_$c.log 2>&1 || { tail -5 $O/pmc_$1_$c.log; exit 1; }
  done
  cd $GRAFT_REPO_ROOT; python3 tools/pmc_reduce.py $O/pmc_$1_FETCH_SIZE $O/pmc_$1_WRITE_SIZE $O --tile $1 --upsample $2 --sources $3 || exit 1; cd /tmp
done
cd $GRAFT_REPO_ROOT
step "SQ counters"; SPX_PAIR64_WAVES=4 bash tools/gpu_sq.sh w4 > $O/sq_w4.log 2>&1 || { tail -5 $O/sq_w4.log; exit 1; }
cp gpurun_out/sq_summary_w4.json $O/; cp gpurun_out/sq_flops_64_u10_w4.json $O/sq_flops_64_u10.json 2>/dev/null
# the traffic / FLOP files of THIS build now exist: put them where bench.py looks (profiles/r03, fingerprint-checked)
# and take the default bench line again, so that the committed line carries traffic and compute_fraction
step "bench again, with this build's counter files"
mv $O/bench.json $O/bench_first.json
cp $O/pmc_traffic*.json $O/sq_flops_64_u10.json profiles/r03/ 2>/dev/null
$B > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
step "phase stamps"; WAVES=4 timeout -k 10 300 python tools/phase_cycles.py > $O/phase_cycles64.txt 2>&1 || tail -3 $O/phase_cycles64.txt
timeout -k 10 300 python tools/phase_cycles128.py > $O/phase_cycles128.txt 2>&1 || tail -3 $O/phase_cycles128.txt
step "rates"; for cfg in "128 20" "96 10" "80 10" "32 10"; do set -- $cfg
  $B --tile $1 --upsample $2 --no-cpu-baseline --no-reference-mode 2>/dev/null > $O/bench_$1_u$2.json || exit 1
done
# the float64 form of the 64 tile's refine (SPX_REFINE_F64): rates next to the default's, and both forms' distance
# from the float64 definition
for cfg in "64 10" "64 20" "80 10"; do set -- $cfg
  $B --tile $1 --upsample $2 --refine float64 --no-cpu-baseline --no-reference-mode 2>/dev/null > $O/bench_$1_u$2_refine_f64.json || exit 1
done
$B --tile 64 --upsample 20 --no-cpu-baseline --no-reference-mode 2>/dev/null > $O/bench_64_u20.json || exit 1
timeout -k 10 500 python tools/refine_precision.py --count 48 --budget 400 2>/dev/null > $O/refine_precision.txt
python tools/bench_disp5.py 2>/dev/null > $O/disp5.txt
python tools/bench_shapes.py 2>/dev/null > $O/shapes.txt
python tools/align_catalog.py 2>/dev/null > $O/align_catalog.txt
python tools/align_synthetic.py 2>/dev/null > $O/align_config5.txt
python bench.py --gpus 2 --one-device --backend gloo --pairs 20000 --steps 5 --warmup 2 > $O/bench_2rank_rehearsal.log 2>&1
step "full test pass"; timeout -k 10 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03e/bench.json'))
r = d['roofline']
print('HEADLINE %.4g pairs/s, kernel %.3f ms, frac %.4f, traffic %s, compute_fraction %s' % (d['value'], r['kernel_ms'], r['frac'], r.get('traffic'), r.get('compute_fraction')))
print('reference mode %.4g displacements/s' % d['reference_mode']['value'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['value_leg'])
PY
