#!/bin/bash
# round 4, call 15: general renderer with the float32 certain-miss test in sphere_hit: parity, throughput
set -u
OUT=gpurun_out/r04_o; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
for rep in 1 2; do timeout -k 10 300 python tools/bench_general.py 256 256 16; done 2>&1 | tee $OUT/bench_general.txt
timeout -k 10 300 python tools/bench_general.py 64 300 100 2>&1 | tee $OUT/bench_general_300.txt
{ echo "## tests/soak_general.py 300 13"; timeout -k 10 600 python tests/soak_general.py 300 13 2>&1 | tail -n 2; } | tee $OUT/soak_general.txt
