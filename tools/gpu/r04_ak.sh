#!/bin/bash
# round 4, call 45: 64 x 12 tiles in the single-shape general kernel for widths they fit better
set -u -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
timeout -k 10 300 python tools/bench_general.py 64 300 100 2>&1 | tee gpurun_out/r04_ak.txt
timeout -k 10 300 python tools/bench_general.py 256 256 16 2>&1 | tee -a gpurun_out/r04_ak.txt
timeout -k 10 600 python tests/soak_general.py 300 29 2>&1 | tail -n 1
