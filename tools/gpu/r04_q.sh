#!/bin/bash
# round 4, call 17: general renderer with the certain-miss test of rectangle_hit too: parity, throughput, soak
set -u
OUT=gpurun_out/r04_q; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest.log
for rep in 1 2; do timeout -k 10 300 python tools/bench_general.py 256 256 16; done 2>&1 | tee $OUT/bench_general.txt
{ echo "## tests/soak_general.py 300 15"; timeout -k 10 600 python tests/soak_general.py 300 15 2>&1 | tail -n 1; } | tee $OUT/soak.txt
