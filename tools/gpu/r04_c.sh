#!/bin/bash
# round 4, call 3: counters of the general renderer's dense pass next to the literal kernel (one_rect, mixed)
set -u
for sc in one_rect mixed; do
  bash profiles/run_profiles.sh r04_general_dense_$sc "256 256 16 --scene $sc" 0 tools/bench_general.py > gpurun_out/r04_c_$sc.log 2>&1
done
export REINFOCUS_GENERAL_DENSE=0
for sc in one_rect; do
  bash profiles/run_profiles.sh r04_general_literal_$sc "256 256 16 --scene $sc" 0 tools/bench_general.py > gpurun_out/r04_c_lit_$sc.log 2>&1
done
tail -3 gpurun_out/r04_c_*.log
