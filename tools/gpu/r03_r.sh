#!/bin/bash
# round 3, GPU call R: final evidence of the headline configuration on the final build
set -u
bash profiles/run_profiles.sh r03 "" 50 > gpurun_out/run_profiles_r03_final.log 2>&1; grep -c "rc=0" gpurun_out/run_profiles_r03_final.log
python bench.py > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err; echo "bench rc=$?"; cut -c1-250 gpurun_out/bench_r03_final.json
