#!/bin/bash
# round 4, call 23: the cooperative single-rectangle general kernel: all general tests, throughput, counters, soak
set -u
OUT=gpurun_out/r04_x; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 3
for rep in 1 2; do timeout -k 10 300 python tools/bench_general.py 256 256 16; done 2>&1 | tee $OUT/bench_general.txt
timeout -k 10 300 python tools/bench_general.py 64 300 100 2>&1 | tee -a $OUT/bench_general.txt
{ echo "## tests/soak_general.py 400 16"; timeout -k 10 600 python tests/soak_general.py 400 16 2>&1 | tail -n 1; } | tee $OUT/soak.txt
bash profiles/run_profiles.sh r04_general_one_rect "256 256 16 --scene one_rect" 0 tools/bench_general.py > gpurun_out/run_profiles_r04_general_one_rect.log 2>&1; echo "profile $(grep -c 'rc=0' gpurun_out/run_profiles_r04_general_one_rect.log)"
