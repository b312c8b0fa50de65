#!/bin/bash
# copy what tools/gpu_round2_profiles.sh left under gpurun_out/ into profiles/r02/ (newest run of each)
O=gpurun_out/r02; P=profiles/r02; mkdir -p $P
cp $O/bench.json $P/bench.json
for f in bench_128_u20 bench_96_u10 bench_80_u10 bench_32_u10 bench_under_rocprofv3 bench_128_under_rocprofv3; do cp $O/$f.json $P/$f.json; done
cp $(ls -t $O/prof64/runc/*_kernel_stats.csv | head -1) $P/bench_kernel_stats.csv
cp $(ls -t $O/prof128/runc/*_kernel_stats.csv | head -1) $P/bench_128_u20_kernel_stats.csv
cp gpurun_out/pmc_traffic_64_u10.json $P/pmc_traffic.json
cp gpurun_out/pmc_traffic_128_u20.json gpurun_out/pmc_traffic_80_u10.json gpurun_out/pmc_traffic_96_u10.json gpurun_out/pmc_traffic_disp5_64.json $P/
for t in 64 128 80 96; do for c in FETCH_SIZE WRITE_SIZE; do cp $(ls -t gpurun_out/pmc_${t}_$c/runc/*_counter_collection.csv | head -1) $P/pmc_${t}_${c}_counter_collection.csv; done; done
for c in FETCH_SIZE WRITE_SIZE; do cp $(ls -t gpurun_out/pmc_disp5_64_$c/*/*_counter_collection.csv | head -1) $P/pmc_disp5_64_${c}_counter_collection.csv; done
cp $(ls -t gpurun_out/pmc_sq/runc/*_counter_collection.csv | head -1) $P/pmc_sq_counter_collection.csv
cp $(ls -t gpurun_out/pmc_sq2/runc/*_counter_collection.csv | head -1) $P/pmc_sq2_counter_collection.csv
cp $O/shapes.txt $O/disp5.txt $O/aux.txt $O/phase_cycles64.txt $O/phase_cycles128.txt $O/usweep.txt $P/
