#!/bin/bash
# round 4, call 5: counters of the dense pass with in-wave loops (one_rect)
set -u
export REINFOCUS_GENERAL_DENSE=2
bash profiles/run_profiles.sh r04_general_dense_inwave_one_rect "256 256 16 --scene one_rect" 0 tools/bench_general.py > gpurun_out/r04_e_one_rect.log 2>&1
tail -n 3 gpurun_out/r04_e_one_rect.log
