#!/bin/bash
# round 4, call 58: the single-shape general kernel only for launches that fill the device
set -u -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
for cfg in "1 300 100" "2 300 100" "16 300 100" "24 300 100" "256 256 16"; do timeout -k 10 300 python tools/bench_general.py $cfg; done 2>&1 | cut -c1-170 | tee gpurun_out/r04_au.txt
timeout -k 10 600 python tests/soak_general.py 300 31 2>&1 | tail -n 1
